"""The ORACLE's per-pixel pipeline against the numpy restatement of SURVEY.md Appendix A (test_oracle_kat._pixel_from_spec)
on seeded random one-brick scenes (CPU only):  python tests/fuzz_spec.py FIRST LAST
Random brick edge (2..12 cells), corner values, camera (outside / inside / along an axis), smooth random TF (the spec is
written in float64 around the quantised filter weight, so only TFs whose neighbouring texels differ little can be held to
2e-5), TF domain, step, opacity scale, gradient shading on / off, frame id (other jitter), frame size; 60 random pixels per case."""
import sys
import time

import numpy as np

from common import Case
from owlexabrick_amd import harness, scenes
from test_oracle_kat import _pixel_from_spec


def check(seed):
    rng = np.random.default_rng(0x5BEC000 + seed)
    n = int(rng.integers(2, 13))
    vals = [float(v) for v in rng.uniform(0, 1, 8)]
    sc = scenes.artificial([[0, 0, 0, n, n, n, 0] + vals], name=f"brick{n}")
    W, H = int(rng.integers(8, 65)), int(rng.integers(8, 49))
    ext = np.array([n, n, n], dtype=np.float64)
    mode = str(rng.choice(["default", "outside", "inside", "axis"]))
    fovy = float(rng.uniform(25.0, 95.0))
    if mode == "default":
        lo, hi = sc.bounds()
        cam = harness.default_camera(lo, hi, W, H)
    else:
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        if mode == "outside":
            o, at = 0.5 * ext + d * float(rng.uniform(0.9, 3.0)) * n, 0.5 * ext + rng.uniform(-0.2, 0.2, 3) * ext
        elif mode == "inside":
            o, at = rng.uniform(0.1, 0.9, 3) * ext, rng.uniform(0, 1, 3) * ext
        else:
            ax = int(rng.integers(0, 3))
            o = 0.5 * ext; o[ax] += float(rng.choice([-1.0, 1.0])) * 1.7 * n
            at = 0.5 * ext
        cam = harness.camera(o, at, [0, 0, 1] if (mode == "axis" and ax == 1) else [0, 1, 0], fovy, W, H)
    xf = harness.default_xf()
    t = np.arange(128) / 127.0
    kind = str(rng.choice(["ramp", "faint", "wave"]))
    if kind == "faint":
        xf[:, 3] = (0.05 * t).astype(np.float32)
    elif kind == "wave":
        xf[:, 3] = (0.5 + 0.45 * np.sin(2 * np.pi * (float(rng.uniform(0.3, 1.5)) * t + float(rng.uniform(0, 1))))).astype(np.float32)
        xf[:, :3] = (0.5 + 0.5 * np.sin(2 * np.pi * (rng.uniform(0.3, 1.2, 3)[None] * t[:, None] + rng.uniform(0, 1, 3)[None]))).astype(np.float32)
    dom = [(0.0, 1.0), (0.2, 0.9), (-1.0, 2.0)][int(rng.integers(0, 3))]
    dt = float(rng.choice([0.5, 0.25, 1.0, 0.37, 2.0]))
    osc = float(rng.choice([1.0, 0.3, 0.05]))
    frame = int(rng.choice([0, 1, 7]))
    grad = int(rng.integers(0, 2))
    case = Case(sc, W=W, H=H, grad=grad, xf=xf, xf_domains=[dom], dt=dt, opacity_scale=osc, frameID=frame, camera=cam)
    rgba, acc, st = case.run_oracle(nthreads=2)
    vol = sc.fields[0].reshape(n, n, n)
    worst, at_px = 0.0, None
    for px, py in zip(rng.integers(0, W, 60), rng.integers(0, H, 60)):
        info = {}
        want = _pixel_from_spec(sc, vol, cam, xf, dom, W, H, int(px), int(py), dt=dt, opacity_scale=osc, frame=frame, grad=bool(grad), info=info)
        if info.get("ill_conditioned"):
            continue                                      # a sample of this ray has a gradient that is rounding noise
        d = float(np.abs(want - acc[py, px, :3]).max())
        if d > worst:
            worst, at_px = d, (int(px), int(py))
    desc = dict(seed=seed, n=n, W=W, H=H, camera=mode, xf=kind, dom=dom, dt=dt, opacity_scale=osc, frame=frame, grad=grad, worst=worst, at=at_px)
    # the spec computes the TF coordinate in float64, the oracle in float32: a sample next to a 1/256 boundary of the
    # filter weight may round to the other side and then moves by one quantisation step of the table (two per pixel allowed)
    step = float(np.abs(np.diff(xf, axis=0)).max()) * osc / 256.0
    tol = 2e-5 + 2.0 * step
    desc["tol"] = tol
    return ([f"pixel {at_px} differs by {worst} > {tol}"] if worst > tol else []), desc


if __name__ == "__main__":
    first, last = int(sys.argv[1]), int(sys.argv[2])
    fails, t0 = 0, time.time()
    for seed in range(first, last + 1):
        bad, desc = check(seed)
        if bad:
            fails += 1
            print(f"FAIL seed {seed}: {desc}", flush=True)
    print(f"{fails} failed of {last - first + 1}, {time.time() - t0:.0f}s")
    sys.exit(1 if fails else 0)
