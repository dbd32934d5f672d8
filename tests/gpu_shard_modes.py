"""per-rank kernel time of rank 0 of N for the wide-march modes: by cost (default), every tile with 2 / 4 lanes per ray, off"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from common import Case
from owlexabrick_amd import scenes
import numpy as np
sc = scenes.config("c4_exajet", scale=1.0)
case = Case(sc, W=2048, H=2048, grad=1, xf_domains=[(0.0, 1.0)])
R = case.hip_renderer()
for world in (2, 4, 8):
    for mode in (1, 0, 2, 4):
        R.setOption("wide_march", mode)
        ms = []
        for rank in (0, world - 1):
            R.setShard(rank, world)
            R.render(); R.render()
            t = []
            for _ in range(3):
                R.render(); t.append(R.stats()["kernel_ms"])
            ms.append(float(np.median(t)))
        print(f"world {world} wide_march {mode}: {max(ms):.3f} ms", flush=True)
