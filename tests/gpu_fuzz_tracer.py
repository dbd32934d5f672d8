"""Seeded random streamline-tracer runs against the oracle:  python tests/gpu_fuzz_tracer.py FIRST LAST
Random three-field scene, seeds (inside, on the boundary, outside the grid), number of traces / timesteps, step length,
channel permutation, camera, both walks; frame by frame as the viewer drives it: traces bit-equal, frames within the
tolerance of tests/common.py."""
import sys
import time

import numpy as np

from common import Case, compare
from fuzz_cases import _random_xf
from owlexabrick_amd import scenes


def check(seed):
    rng = np.random.default_rng(0x7AC3000 + seed)
    B = int(rng.choice([2, 4, 4]))
    levels = int(rng.integers(1, 4))
    root = tuple(int(v) for v in rng.integers(1, 4, 3))
    sc = scenes.amr(seed=int(rng.integers(1, 1 << 20)), root=root, B=B, levels=levels, feature=str(rng.choice(["shell", "plume"])), fields=3)
    ext = np.array(root, dtype=np.float64) * B * (1 << (levels - 1))
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    camera = (list(0.5 * ext + d * float(rng.uniform(1.0, 2.2)) * ext.max()), list(0.5 * ext), [0, 1, 0], float(rng.uniform(35, 80)))
    accel = int(rng.integers(0, 2))
    case = Case(sc, W=int(rng.integers(32, 97)), H=int(rng.integers(24, 81)), grad=int(rng.integers(0, 2)),
                opacity_scale=float(rng.choice([0.02, 0.3, 1.0])), accel=accel, fast_math=0, camera=camera,
                xf=[_random_xf(rng, str(rng.choice(["ramp", "band", "faint"])))] * 3, xf_domains=[(0.0, 1.0)] * 3,
                iso=[(float(rng.uniform(0.2, 0.8)), 0)] if rng.uniform() < 0.3 else None)
    n = int(rng.integers(1, 60))
    seeds = rng.uniform(-0.15, 1.15, size=(n, 3)) * ext                  # some start outside the grid
    if n > 3:
        seeds[0] = 0.0; seeds[1] = ext; seeds[2, 0] = ext[0]              # corners / faces of the domain
    seeds = seeds.astype(np.float32)
    nt = int(rng.integers(2, 9))
    steplen = float(rng.choice([0.5, 2.0, 6.0, 20.0]))
    chans = tuple(int(c) for c in rng.permutation(3))
    desc = dict(seed=seed, B=B, levels=levels, root=root, accel=accel, n=n, nt=nt, steplen=steplen, chans=chans)
    S = case.oracle_scene()
    S.reset_tracer(True, chans, n, nt, steplen, seeds)
    R = case.hip_renderer()
    R.resetTracer(seeds, channels=chans, numTimesteps=nt, steplen=steplen, enabled=True)
    bad = []
    for f in range(nt + 1):
        S.advance_tracer()
        R.advanceTracer()
        fs, P = case.oracle_state(S)
        o = S.render(fs, P, case.W, case.H, nthreads=8)
        R.updateFrameID(0)
        h_rgba, h_acc = R.render(), R.readAccum()
        if not np.array_equal(R.readTraces(), S.traces()):
            bad.append(f"traces differ at frame {f}")
            break
        r = compare(o, (h_rgba, h_acc, None))
        if (r["accum_bad"] or r["rgba_bad"]) and not (r["flip_pixels"] > 0 and r["flips_ok"] and r["rgba_bad"] <= 3 * r["flip_pixels"]):
            bad.append(f"frame {f}: {r}")
            break
    R.close()
    return bad, desc


if __name__ == "__main__":
    first, last = int(sys.argv[1]), int(sys.argv[2])
    fails, t0 = 0, time.time()
    for seed in range(first, last + 1):
        bad, desc = check(seed)
        if bad:
            fails += 1
            print(f"FAIL seed {seed}: {desc} {bad}", flush=True)
        elif seed % 10 == 0:
            print(f"seed {seed} ok ({time.time() - t0:.0f}s)", flush=True)
    print(f"{fails} failed of {last - first + 1}, {time.time() - t0:.0f}s", flush=True)
    sys.exit(1 if fails else 0)
