import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from common import Case
from owlexabrick_amd import scenes
for ex in ("ex4", "ex3"):
  for grad in (0, 1):
    for gi in (0, 1):
      for iso in (None, [(0.5, 0)]):
        c = Case(scenes.example(ex), W=96, H=64, grad=grad, iso=iso, grad_iso=gi, basis_form=1, fast_math=0, accel=1)
        o = c.run_oracle()[2]; h = c.run_hip(stats=True)[2]
        print(ex, "grad", grad, "grad_iso", gi, "iso", iso, {k: (o[k], h[k]) for k in ("brick_visits", "corner_loads", "iso_evals")}, flush=True)
