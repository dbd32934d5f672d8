"""The ORACLE's implicit iso-surface against the numpy restatement test_oracle_kat._iso_pixel_from_spec on seeded random
one-brick scenes (CPU only):  python tests/fuzz_spec_iso.py FIRST LAST
Random brick edge, corner values, camera, iso value, smooth random colour table with alpha 0 (the pixel is then the shaded
surface), iso gradient shading on / off, AO rays (infinite / finite length), step, frame id; 60 random pixels per case.  Rays with an ill-conditioned gradient
sample (flat field) or a filter weight on a rounding tie are skipped."""
import sys
import time

import numpy as np

from common import Case
from owlexabrick_amd import harness, scenes
from test_oracle_kat import _iso_pixel_from_spec


def build(seed):
    """everything check() needs, also for debugging one pixel by hand"""
    rng = np.random.default_rng(0x150C000 + seed)
    n = int(rng.integers(2, 11))
    vals = [float(v) for v in rng.uniform(0, 1, 8)]
    sc = scenes.artificial([[0, 0, 0, n, n, n, 0] + vals], name=f"brick{n}")
    W, H = int(rng.integers(8, 65)), int(rng.integers(8, 49))
    ext = np.array([n, n, n], dtype=np.float64)
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    if rng.uniform() < 0.7:
        o, at = 0.5 * ext + d * float(rng.uniform(1.0, 3.0)) * n, 0.5 * ext + rng.uniform(-0.2, 0.2, 3) * ext
    else:
        o, at = rng.uniform(0.1, 0.9, 3) * ext, rng.uniform(0, 1, 3) * ext
    cam = harness.camera(o, at, [0, 1, 0], float(rng.uniform(25.0, 90.0)), W, H)
    xf = harness.default_xf()
    t = np.arange(128) / 127.0
    xf[:, 3] = 0.0
    xf[:, :3] = (0.5 + 0.45 * np.sin(2 * np.pi * (rng.uniform(0.2, 1.0, 3)[None] * t[:, None] + rng.uniform(0, 1, 3)[None]))).astype(np.float32)
    vol = sc.fields[0].reshape(n, n, n)
    iso = float(rng.uniform(float(vol.min()) + 0.05 * float(np.ptp(vol)), float(vol.max()) - 0.05 * float(np.ptp(vol))))
    grad_iso = int(rng.integers(0, 2))
    dt = float(rng.choice([0.5, 0.25, 1.0, 0.37]))
    frame = int(rng.choice([0, 1, 5]))
    ao_length = [None, None, 1e20, 0.4 * n][int(rng.integers(0, 4))] if grad_iso else None      # AO needs a shaded hit
    case = Case(sc, W=W, H=H, grad=int(rng.integers(0, 2)), grad_iso=grad_iso, iso=[(iso, 0)], xf=xf, xf_domains=[(0.0, 1.0)],
                dt=dt, frameID=frame, camera=cam, ao=int(ao_length is not None), ao_length=(ao_length or 1e20))
    return dict(rng=rng, case=case, vol=vol, cam=cam, xf=xf, W=W, H=H, iso=iso, grad_iso=grad_iso, dt=dt, frame=frame, ao_length=ao_length, n=n)


def check(seed):
    b = build(seed)
    rng, case, vol, cam, xf, W, H, iso, grad_iso, dt, frame, ao_length, n = (b[k] for k in ("rng", "case", "vol", "cam", "xf", "W", "H", "iso", "grad_iso", "dt", "frame", "ao_length", "n"))
    rgba, acc, st = case.run_oracle(nthreads=2)
    step = float(np.abs(np.diff(xf[:, :3], axis=0)).max()) / 256.0
    tol = 3e-5 + 2.0 * step
    worst, at_px, compared, hits = 0.0, None, 0, 0
    for px, py in zip(rng.integers(0, W, 60), rng.integers(0, H, 60)):
        info = {}
        want = _iso_pixel_from_spec(vol, cam, xf, (0.0, 1.0), W, H, int(px), int(py), iso, bool(grad_iso), dt=dt, frame=frame, info=info, ao_length=ao_length)
        if info:
            continue
        compared += 1
        hits += int(want.sum() > 0)
        dd = float(np.abs(want - acc[py, px, :3]).max())
        if dd > worst:
            worst, at_px = dd, (int(px), int(py))
    desc = dict(seed=seed, n=n, W=W, H=H, iso=iso, grad_iso=grad_iso, ao=ao_length, dt=dt, frame=frame, compared=compared, hits=hits, worst=worst, at=at_px, tol=tol,
                iso_segments=st["iso_segments"])
    return ([f"pixel {at_px} differs by {worst} > {tol}"] if worst > tol else []), desc


if __name__ == "__main__":
    first, last = int(sys.argv[1]), int(sys.argv[2])
    fails, hits, t0 = 0, 0, time.time()
    for seed in range(first, last + 1):
        bad, desc = check(seed)
        hits += desc["hits"]
        if bad:
            fails += 1
            print(f"FAIL seed {seed}: {desc}", flush=True)
    print(f"{fails} failed of {last - first + 1} ({hits} surface pixels compared), {time.time() - t0:.0f}s")
    sys.exit(1 if fails else 0)
