"""The ORACLE's region loop against the numpy restatement test_oracle_kat._pixel_from_spec_regions on seeded random
MULTI-region scenes (CPU only):  python tests/fuzz_spec_regions.py FIRST LAST
Random partitions into bricks of any shape and level (holes, level jumps), space skipping off, smooth TF, gradient shading
off (the un-normalised gradient of a level boundary is ill-conditioned where the field is flat; the gradient itself is
checked by the hat-basis sweep), random camera / step / opacity scale / frame id / clip box / a second primary channel; 24 random pixels per case."""
import sys
import time

import numpy as np

from common import Case, po
from fuzz_cases import _random_grids
from owlexabrick_amd import harness, scenes
from test_oracle_kat import _pixel_from_spec_regions


def check(seed):
    rng = np.random.default_rng(0x4E610000 + seed)
    grids, ext = _random_grids(rng)
    sc = scenes.artificial(grids, name=f"grids{seed}")
    W, H = int(rng.integers(8, 49)), int(rng.integers(8, 41))
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    if rng.uniform() < 0.75:
        o, at = 0.5 * ext + d * float(rng.uniform(0.9, 2.5)) * ext.max(), 0.5 * ext + rng.uniform(-0.2, 0.2, 3) * ext
    else:
        o, at = rng.uniform(0.1, 0.9, 3) * ext, rng.uniform(0, 1, 3) * ext
    cam = harness.camera(o, at, [0, 1, 0], float(rng.uniform(25.0, 90.0)), W, H)
    xf = harness.default_xf()
    t = np.arange(128) / 127.0
    kind = str(rng.choice(["ramp", "faint", "wave"]))
    if kind == "faint":
        xf[:, 3] = (0.05 * t).astype(np.float32)
    elif kind == "wave":
        xf[:, 3] = (0.5 + 0.45 * np.sin(2 * np.pi * (float(rng.uniform(0.3, 1.2)) * t + float(rng.uniform(0, 1))))).astype(np.float32)
    dt = float(rng.choice([0.5, 0.25, 1.0, 0.37]))
    osc = float(rng.choice([1.0, 0.3, 0.05]))
    frame = int(rng.choice([0, 2]))
    clip = None
    if rng.uniform() < 0.3:
        clo = rng.uniform(0.0, 0.4, 3) * ext
        clip = (list(clo), list(clo + rng.uniform(0.3, 0.6, 3) * ext))
    more = []
    if rng.uniform() < 0.35:                                    # a second primary channel with its own TF and domain
        k = rng.uniform(0.05, 0.4, 3)
        scenes.with_extra_field(sc, lambda ctr: 0.5 + 0.5 * np.sin(ctr @ k))
        xf2 = harness.default_xf()
        xf2[:, 3] = (0.3 * (1.0 - t)).astype(np.float32)
        xf2[:, :3] = xf2[:, ::-1][:, 1:4]
        more.append((xf2, (0.1, 0.9)))
    case = Case(sc, W=W, H=H, grad=0, xf=[xf] + [m[0] for m in more], xf_domains=[(0.0, 1.0)] + [m[1] for m in more], dt=dt,
                opacity_scale=osc, frameID=frame, camera=cam, space_skipping=0, clip=clip)
    rgba, acc, st = case.run_oracle(nthreads=2)
    S = po.OracleScene(sc.bricks7, sc.cellIDs, sc.fields)
    tol = 3e-5 + 2.0 * max(float(np.abs(np.diff(x_, axis=0)).max()) for x_ in [xf] + [m[0] for m in more]) * osc / 256.0
    worst, at_px, compared, lit = 0.0, None, 0, 0
    for px, py in zip(rng.integers(0, W, 24), rng.integers(0, H, 24)):
        info = {}
        want = _pixel_from_spec_regions(S, cam, xf, (0.0, 1.0), W, H, int(px), int(py), dt=dt, opacity_scale=osc, frame=frame, info=info, clip=clip,
                                        more_channels=more)
        if info:
            continue
        compared += 1
        lit += int(want.sum() > 0)
        dd = float(np.abs(want - acc[py, px, :3]).max())
        if dd > worst:
            worst, at_px = dd, (int(px), int(py))
    desc = dict(seed=seed, bricks=len(grids), regions=int(S.num_regions), W=W, H=H, xf=kind, dt=dt, opacity_scale=osc, frame=frame,
                compared=compared, lit=lit, worst=worst, at=at_px, tol=tol)
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
    print(f"{fails} failed of {last - first + 1} ({lit} lit pixels compared), {time.time() - t0:.0f}s")
    sys.exit(1 if fails else 0)
