import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from common import Case
from owlexabrick_amd import scenes
import numpy as np
W, H = 24, 16
case = Case(scenes.example("ex4"), W=W, H=H, iso=[(0.5, 0)], grad_iso=0)
R = case.hip_renderer()
keys = ["segments", "sample_evals", "samples", "brick_visits", "corner_loads", "iso_segments", "iso_evals"]
n = 0
for py in range(H):
    for px in range(W):
        R.setOption("debug_pixel", px + W * py)
        _, st = R.renderStats()
        o = case.run_oracle(nthreads=1, window=(px, py, px + 1, py + 1))[2]
        if any(o[k] != st[k] for k in keys):
            print((px, py), {k: (o[k], st[k]) for k in keys})
            n += 1
            if n > 5: sys.exit(0)
print("done", n)
