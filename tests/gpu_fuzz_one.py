"""debug helper: one rich seed, kd vs LBVH vs oracle, where they differ"""
import sys
import numpy as np
from common import compare
from fuzz_cases import random_rich_case, random_case

seed = int(sys.argv[1]); rich = "--rich" in sys.argv
if rich:
    case, desc, frames = random_rich_case(seed)
else:
    (case, desc), frames = random_case(seed), 1
print(desc)
case.fast_math = 0
o = case.run_oracle(frames=frames)
print("oracle", {k: o[2][k] for k in ("segments", "samples", "iso_segments", "iso_evals")})
outs = {}
for accel in (1, 0):
    case.accel = accel
    h = case.run_hip(stats=True, frames=frames)
    outs[accel] = h
    print("accel", accel, {k: h[2][k] for k in ("segments", "samples", "iso_segments", "iso_evals")}, compare(o, h))
d = np.abs(outs[1][1] - outs[0][1]).max(axis=-1)
ys, xs = np.nonzero(d > 1e-4)
print("kd vs lbvh differing pixels:", len(ys), list(zip(xs[:10].tolist(), ys[:10].tolist())))
do = np.abs(outs[1][1] - o[1]).max(axis=-1)
ys, xs = np.nonzero(do > 1e-4)
print("kd vs oracle differing pixels:", len(ys), list(zip(xs[:10].tolist(), ys[:10].tolist())))
for x, y in list(zip(xs[:5].tolist(), ys[:5].tolist())):
    print((x, y), "oracle", o[1][y, x], "kd", outs[1][1][y, x], "lbvh", outs[0][1][y, x])
