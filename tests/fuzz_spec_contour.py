"""The ORACLE's contour planes against the geometric restatement test_oracle_kat._contour_pixel_from_spec (CPU only):
    python tests/fuzz_spec_contour.py FIRST LAST
Random brick partitions, one to three planes with random (not normalised) normals and offsets, smooth colour table with
alpha 0, random camera and frame id; 120 random pixels per case (pixels at a polygon edge, at a region face or where two
planes coincide are left out)."""
import sys
import time

import numpy as np

from common import Case, po
from fuzz_cases import _random_grids
from owlexabrick_amd import harness, scenes
from test_oracle_kat import _contour_pixel_from_spec


def check(seed):
    rng = np.random.default_rng(0xC0470000 + seed)
    grids, ext = _random_grids(rng)
    sc = scenes.artificial(grids, name=f"grids{seed}")
    W, H = int(rng.integers(8, 49)), int(rng.integers(8, 41))
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    o, at = 0.5 * ext + d * float(rng.uniform(0.9, 2.5)) * ext.max(), 0.5 * ext + rng.uniform(-0.2, 0.2, 3) * ext
    cam = harness.camera(o, at, [0, 1, 0], float(rng.uniform(25.0, 90.0)), W, H)
    xf = harness.default_xf()
    t = np.arange(128) / 127.0
    xf[:, 3] = 0.0
    xf[:, :3] = (0.5 + 0.45 * np.sin(2 * np.pi * (rng.uniform(0.2, 1.0, 3)[None] * t[:, None] + rng.uniform(0, 1, 3)[None]))).astype(np.float32)
    planes = [([float(v) for v in rng.normal(size=3) * float(rng.choice([1.0, 0.5, 2.0]))], float(rng.uniform(-0.3, 0.9)), 0)
              for _ in range(int(rng.integers(1, 4)))]
    frame = int(rng.choice([0, 3]))
    case = Case(sc, W=W, H=H, grad=int(rng.integers(0, 2)), xf=xf, xf_domains=[(0.0, 1.0)], contour=planes, frameID=frame, camera=cam)
    rgba, acc, st = case.run_oracle(nthreads=2)
    S = po.OracleScene(sc.bricks7, sc.cellIDs, sc.fields)
    tol = 3e-5 + 2.0 * float(np.abs(np.diff(xf[:, :3], axis=0)).max()) / 256.0 * 3.0
    worst, at_px, compared, lit = 0.0, None, 0, 0
    for px, py in zip(rng.integers(0, W, 120), rng.integers(0, H, 120)):
        info = {}
        want = _contour_pixel_from_spec(S, cam, xf, (0.0, 1.0), W, H, int(px), int(py), planes, frame=frame, info=info)
        if info:
            continue
        compared += 1
        lit += int(want.sum() > 0)
        dd = float(np.abs(want - acc[py, px, :3]).max())
        if dd > worst:
            worst, at_px = dd, (int(px), int(py))
    desc = dict(seed=seed, bricks=len(grids), planes=len(planes), W=W, H=H, frame=frame, compared=compared, lit=lit, worst=worst, at=at_px, tol=tol)
    return ([f"pixel {at_px} differs by {worst} > {tol}"] if worst > tol else []), desc


if __name__ == "__main__":
    first, last = int(sys.argv[1]), int(sys.argv[2])
    fails, lit, t0 = 0, 0, time.time()
    for seed in range(first, last + 1):
        bad, desc = check(seed)
        lit += desc["lit"]
        if bad:
            fails += 1
            print(f"FAIL seed {seed}: {desc}", flush=True)
    print(f"{fails} failed of {last - first + 1} ({lit} plane pixels compared), {time.time() - t0:.0f}s")
    sys.exit(1 if fails else 0)
