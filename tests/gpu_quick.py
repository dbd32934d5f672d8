"""ad-hoc GPU parity sweep (run on the GPU box): python tests/gpu_quick.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from common import Case, compare, band_xf
from owlexabrick_amd import scenes
import numpy as np

cases = []
for name in ["ex0", "ex1", "ex2", "ex3", "ex4"]:
    cases.append((name, Case(scenes.example(name), W=96, H=64, grad=0)))
    cases.append((name + "+grad", Case(scenes.example(name), W=96, H=64, grad=1)))
cases.append(("ex3+iso", Case(scenes.example("ex3"), W=96, H=64, grad=1, iso=[(0.4, 0)])))
cases.append(("ex4+iso+ao", Case(scenes.example("ex4"), W=96, H=64, grad=0, iso=[(0.5, 0)], ao=1, ao_length=3.0)))
cases.append(("c1_64", Case(scenes.example("c1_64"), W=128, H=128, grad=0)))
cases.append(("c1_64+grad+iso", Case(scenes.example("c1_64"), W=128, H=128, grad=1, iso=[(0.3, 0)])))
amr = scenes.amr(seed=3, root=(3, 3, 2), B=4, levels=3)
cases.append(("amr", Case(amr, W=128, H=128, grad=0)))
cases.append(("amr+grad", Case(amr, W=128, H=128, grad=1)))
cases.append(("amr+band", Case(amr, W=128, H=128, grad=0, xf=band_xf())))
cases.append(("amr+noskip", Case(amr, W=128, H=128, grad=0, xf=band_xf(), space_skipping=0)))
cases.append(("amr+iso", Case(amr, W=128, H=128, grad=1, iso=[(0.45, 0)])))
amr2 = scenes.amr(seed=5, root=(2, 2, 2), B=4, levels=3, feature="plume", fields=2)
cases.append(("amr2ch", Case(amr2, W=128, H=128, grad=1)))
for name, c in cases:
    t = time.time(); o = c.run_oracle(); to = time.time() - t
    t = time.time(); h = c.run_hip(stats=True); th = time.time() - t
    r = compare(o, h, name)
    st_o, st_h = o[2], h[2]
    keys = ["segments", "sample_evals", "samples", "brick_visits", "corner_loads", "iso_segments", "iso_evals"]
    same = all(st_o[k] == st_h[k] for k in keys)
    print(f"{name:16s} {r} stats_equal={same} kernel_ms={st_h['kernel_ms']:.3f} nodes={st_h['nodes_visited']} oracle_s={to:.2f}", flush=True)
    if not same:
        print("   oracle", {k: st_o[k] for k in keys}); print("   hip   ", {k: st_h[k] for k in keys})
