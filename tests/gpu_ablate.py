import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from common import Case
from owlexabrick_amd import scenes
sc = scenes.config("c4_exajet", scale=1.0)
case = Case(sc, W=2048, H=2048, grad=1, xf_domains=[(0.0, 1.0)])
R = case.hip_renderer()
for ab in (0, 1, 2, 3, 0):
    R.setOption("ablate", ab)
    R.render(); R.render()
    t = []
    for k in range(3):
        R.render(); t.append(R.stats()["kernel_ms"])
    print("ablate", ab, "kernel_ms", min(t))
