"""Which walk for which transfer function (GPU box): python tests/gpu_walk_choice.py [config] [size]
The stack walk skips subtrees without an active region, the rope walk passes through every leaf on the ray.  For transfer
functions that leave different fractions of the regions active: kernel time of the DVR march with either walk, and what the
module's automatic choice (option walk = 0) takes.  Frames are bit-identical between the walks (checked)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from common import Case
from owlexabrick_amd import harness, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "c4_exajet"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
sc = scenes.config(name)
holes = float(os.environ.get("EXA_WALK_HOLES", "0"))          # > 0: that fraction of the cells missing (the ALLOW_EMPTY_CELLS kernels)
if holes > 0:
    sc = scenes.with_empty_cells(sc, fraction=holes, seed=1)
case = Case(sc, W=size, H=size, grad=1, xf_domains=[(0.0, 1.0)] * len(sc.fields), allow_empty_cells=holes > 0)
R = case.hip_renderer()
t = np.arange(128) / 127.0


def tf(alpha):
    xf = harness.default_xf()
    xf[:, 3] = alpha
    return xf


TFS = {"ramp (reference default)": tf(t),
       "band 0.35-0.65": tf(np.where((t >= 0.35) & (t <= 0.65), 0.6, 0.0)),
       "band 0.45-0.55": tf(np.where((t >= 0.45) & (t <= 0.55), 0.6, 0.0)),
       "upper half": tf(np.where(t >= 0.5, 0.5, 0.0)),
       "upper quarter": tf(np.where(t >= 0.75, 0.5, 0.0)),
       "top 10 %": tf(np.where(t >= 0.9, 0.8, 0.0)),
       "lower quarter": tf(np.where(t <= 0.25, 0.3, 0.0)),
       # thin TFs: little opacity anywhere, so that rays are long and the walk is most of the frame
       "below 0.05, thin": tf(np.where(t <= 0.05, 0.02, 0.0)),
       "below 0.1, thin": tf(np.where(t <= 0.1, 0.02, 0.0)),
       "below 0.15, thin": tf(np.where(t <= 0.15, 0.02, 0.0)),
       "below 0.25, thin": tf(np.where(t <= 0.25, 0.02, 0.0)),
       "above 0.2, thin": tf(np.where(t >= 0.2, 0.02, 0.0)),
       "above 0.3, thin": tf(np.where(t >= 0.3, 0.02, 0.0)),
       "above 0.4, thin": tf(np.where(t >= 0.4, 0.02, 0.0)),
       "everything, thin": tf(np.full(128, 0.02))}
for label, xf in TFS.items():
    for c in range(len(sc.fields)):
        R.updateXF(c, xf[:, 3], xf[:, :3], (0.0, 1.0), 1.0)
    act = R.readActivity(0)
    frac = float(act.mean())
    out, ms = {}, {}
    for walk in (1, 2, 0):
        R.setOption("walk", walk)
        for _ in range(3):                       # cost feedback: measure, re-order, steady
            R.render()
        ks = []
        for _ in range(5):
            rgba = R.render()
            ks.append(R.stats()["kernel_ms"])
        ms[walk] = float(np.mean(ks))
        out[walk] = (rgba.copy(), R.readAccum().copy())
    same = all(np.array_equal(out[1][0], out[w][0]) and np.array_equal(out[1][1].view(np.uint32), out[w][1].view(np.uint32)) for w in (2, 0))
    print(f"{label:26s} active {frac:6.3f}  stack {ms[1]:7.3f} ms  rope {ms[2]:7.3f} ms  auto {ms[0]:7.3f} ms  identical={same}", flush=True)
R.close()
