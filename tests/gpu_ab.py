"""A/B timing of runtime options on the bench scene, interleaved in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from common import Case
from owlexabrick_amd import scenes
import numpy as np
key = sys.argv[1]; vals = [int(v) for v in sys.argv[2:]]
sc = scenes.config("c4_exajet", scale=1.0)
case = Case(sc, W=2048, H=2048, grad=1, xf_domains=[(0.0, 1.0)])
R = case.hip_renderer()
res = {v: [] for v in vals}
for rnd in range(4):
    for v in vals:
        R.setOption(key, v)
        if key == "tile_order": R.resizeFrameBuffer((2048, 2048))
        R.render()
        R.render(); res[v].append(R.stats()["kernel_ms"])
for v in vals: print(key, v, "min %.3f med %.3f" % (min(res[v]), float(np.median(res[v]))))
