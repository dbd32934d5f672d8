"""The ORACLE's triangle surfaces against the geometric restatement test_oracle_kat._mesh_pixel_from_spec (CPU only):
    python tests/fuzz_spec_mesh.py FIRST LAST
Random triangle soups (also degenerate slivers and triangles behind the camera) over a transparent volume; 150 random
pixels per case (pixels on a triangle edge or with two triangles at one distance are left out)."""
import sys
import time

import numpy as np

from common import Case
from owlexabrick_amd import harness, scenes
from test_oracle_kat import _mesh_pixel_from_spec


def check(seed):
    rng = np.random.default_rng(0x3E5A0000 + seed)
    sc = scenes.example("ex2")                                  # 8^3 brick; the volume stays transparent
    ext = np.array([8.0, 8.0, 8.0])
    W, H = int(rng.integers(8, 65)), int(rng.integers(8, 49))
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    o = 0.5 * ext + d * float(rng.uniform(0.3, 2.5)) * 8.0      # sometimes among the triangles
    cam = harness.camera(o, 0.5 * ext + rng.uniform(-0.2, 0.2, 3) * ext, [0, 1, 0], float(rng.uniform(25.0, 90.0)), W, H)
    nt = int(rng.integers(1, 12))
    verts = (rng.uniform(-0.3, 1.3, size=(3 * nt, 3)) * ext).astype(np.float32)
    if rng.uniform() < 0.3:
        verts[2] = verts[0] + 1e-3 * (verts[1] - verts[0])       # a sliver
    tris = np.arange(3 * nt, dtype=np.int32).reshape(nt, 3)
    xf = harness.default_xf(); xf[:, 3] = 0.0
    frame = int(rng.choice([0, 2]))
    case = Case(sc, W=W, H=H, xf=xf, xf_domains=[(0.0, 1.0)], meshes=[(verts, tris)], frameID=frame, camera=cam)
    rgba, acc, st = case.run_oracle(nthreads=2)
    worst, at_px, compared, lit = 0.0, None, 0, 0
    for px, py in zip(rng.integers(0, W, 150), rng.integers(0, H, 150)):
        info = {}
        want = _mesh_pixel_from_spec(cam, W, H, int(px), int(py), verts, tris, frame=frame, info=info)
        if info:
            continue
        compared += 1
        lit += int(want.sum() > 0)
        dd = float(np.abs(want - acc[py, px, :3]).max())
        if dd > worst:
            worst, at_px = dd, (int(px), int(py))
    desc = dict(seed=seed, tris=nt, W=W, H=H, frame=frame, compared=compared, lit=lit, worst=worst, at=at_px)
    return ([f"pixel {at_px} differs by {worst}"] if worst > 3e-5 else []), desc


if __name__ == "__main__":
    first, last = int(sys.argv[1]), int(sys.argv[2])
    fails, lit, t0 = 0, 0, time.time()
    for seed in range(first, last + 1):
        bad, desc = check(seed)
        lit += desc["lit"]
        if bad:
            fails += 1
            print(f"FAIL seed {seed}: {desc}", flush=True)
    print(f"{fails} failed of {last - first + 1} ({lit} triangle pixels compared), {time.time() - t0:.0f}s")
    sys.exit(1 if fails else 0)
