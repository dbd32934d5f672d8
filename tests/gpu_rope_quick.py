"""ad-hoc check of the rope walk on the GPU box: python tests/gpu_rope_quick.py
every case of gpu_quick.py with the stack walk and the rope walk against the oracle (counting variant, library powf) and
the shipped rope kernel against the counting one"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from common import Case, compare, band_xf
from owlexabrick_amd import scenes

KEYS = ["segments", "sample_evals", "samples", "brick_visits", "corner_loads", "iso_segments", "iso_evals"]
cases = []
for name in ["ex0", "ex1", "ex2", "ex3", "ex4"]:
    cases.append((name + "+grad", lambda name=name: Case(scenes.example(name), W=96, H=64, grad=1)))
cases.append(("ex3+iso", lambda: Case(scenes.example("ex3"), W=96, H=64, grad=1, iso=[(0.4, 0)])))
cases.append(("c1_64", lambda: Case(scenes.example("c1_64"), W=128, H=128, grad=0)))
amr = scenes.amr(seed=3, root=(3, 3, 2), B=4, levels=3)
cases.append(("amr", lambda: Case(amr, W=128, H=128, grad=0)))
cases.append(("amr+grad", lambda: Case(amr, W=128, H=128, grad=1)))
cases.append(("amr+band", lambda: Case(amr, W=128, H=128, grad=0, xf=band_xf())))
cases.append(("amr+noskip", lambda: Case(amr, W=128, H=128, grad=0, xf=band_xf(), space_skipping=0)))
cases.append(("amr+iso", lambda: Case(amr, W=128, H=128, grad=1, iso=[(0.45, 0)])))
cases.append(("amr+inside", lambda: Case(amr, W=96, H=96, grad=1, camera=([20.3, 22.1, 14.2], [30, 20, 10], [0, 1, 0], 80.0))))
cases.append(("amr+clip", lambda: Case(amr, W=96, H=96, grad=1, clip=([10, 8, 4], [40, 40, 28]))))
amr2 = scenes.amr(seed=5, root=(2, 2, 2), B=4, levels=3, feature="plume", fields=2)
cases.append(("amr2ch", lambda: Case(amr2, W=128, H=128, grad=1)))
cases.append(("gen_exajet", lambda: Case(scenes.generated(kind="exajet", seed=11, root=(4, 2, 2), B=8, levels=3), W=160, H=96, grad=1)))
bad = 0
for name, mk in cases:
    c = mk()
    c.fast_math = 0
    o = c.run_oracle()
    for accel in (1, 2):
        c.accel = accel
        h = c.run_hip(stats=True)
        r = compare(o, h, name)
        same = all(o[2][k] == h[2][k] for k in KEYS)
        ok = same and r["accum_bad"] == 0 and r["rgba_bad"] == 0 and h[2]["diag"][8] == 0
        bad += not ok
        print(f"{name:12s} walk {accel} ok={ok} accum_max={r['accum_max']:.3g} bad={r['accum_bad']} stats_equal={same} mismatches={h[2]['diag'][8]} "
              f"nodes16={h[2]['nodes_visited']} leaves={h[2]['walk_leaf_visits']} segs={h[2]['segments']} restarts={h[2]['walk_restarts']}", flush=True)
        if not same:
            print("   oracle", {k: o[2][k] for k in KEYS}); print("   hip   ", {k: h[2][k] for k in KEYS})
    # shipped rope kernel (fast_math default) == its counting variant
    c.fast_math = None
    plain, counted = c.run_hip(), c.run_hip(stats=True)
    eq = np.array_equal(plain[1].view(np.uint32), counted[1].view(np.uint32))
    bad += not eq
    print(f"{name:12s} shipped rope kernel == counting variant: {eq}", flush=True)
print("FAILURES", bad)
sys.exit(1 if bad else 0)
