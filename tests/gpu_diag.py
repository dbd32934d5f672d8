"""phase/lane diagnostics of the v2 kernel on the bench scene (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from common import Case
from owlexabrick_amd import scenes, binding, harness
import numpy as np
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
sc = scenes.config("c4_exajet", scale=scale)
case = Case(sc, W=2048, H=2048, grad=1, xf_domains=[(0.0, 1.0)])
R = case.hip_renderer()
if os.environ.get("EXA_DIAG_WALK"):                 # 1 = stack walk, 2 = rope walk (default: the module's choice)
    R.setOption("walk", int(os.environ["EXA_DIAG_WALK"]))
_, st = R.renderStats()
d = st["diag"]
names = ["brick", "final", "node", "leaf"]
print({k: v for k, v in st.items() if k not in ("diag", "phase_cycles")})
for i, n in enumerate(names):
    w, l = d[2 * i], d[2 * i + 1]
    print(f"{n:6s} wave-execs {w:.4g} lanes {l:.4g} util {l / max(1, 64 * w):.3f}")
print("kd mismatches", d[8])
if st["walk_leaf_visits"]:
    inner = st["nodes_visited"] - 4 * st["walk_leaf_visits"]
    print(f"rope walk: {st['walk_leaf_visits']} leaves + {inner} inner nodes = {(st['walk_leaf_visits'] + inner) / st['segments']:.3f} visits per segment")
else:
    print(f"stack walk: {st['nodes_visited'] / st['segments']:.3f} node visits per segment, {st['walk_restarts']} restarts")
R.setOption("stats_mode", 2)
_, st2 = R.renderStats()
R.setOption("stats_mode", 1)
print("timing variant kernel_ms", st2["kernel_ms"])
pc = st2["phase_cycles"]
tot = float(sum(pc)) or 1.0
for n, c in zip(["brick visit", "sample epilogue", "kd walk", "segment pop", "other"], pc):
    print(f"wave cycles in {n:16s} {c:.4g}  {100 * c / tot:.1f} %")
for k in range(3):
    R.render()
print("kernel_ms", R.stats()["kernel_ms"])
