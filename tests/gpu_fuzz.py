"""Sweep of seeded random parity cases on a GPU box (not collected by pytest):
    python tests/gpu_fuzz.py FIRST LAST [--rich | --grids | --many | --deep | --domains] [--clip] [--holes] [--keep-going]
For each seed: the oracle against the HIP module through the C ABI, both walks (kd, LBVH), library powf
(`fast_math = 0`): accumulation buffer within the tolerance of tests/common.py, RGBA8 within 1 LSB, identical work
counters, no slab-test mismatch; and the shipped kernel (counting off, `fast_math` 0 and 1 defaults) equal to the counting
variant bit for bit / within the flip tolerance."""
import sys
import time

import numpy as np

from common import compare
from common import ACCUM_ATOL, FLIP_BOUND
from fuzz_cases import random_case, random_deep_case, random_rich_case

STAT_KEYS = ["segments", "sample_evals", "samples", "brick_visits", "corner_loads", "iso_segments", "iso_evals"]


def check(seed, rich=False, clipbox=False, holes=False):
    """list of failure strings (empty = pass)"""
    bad = []
    frames = 1
    if rich is True:
        case, desc, frames = random_rich_case(seed)
    elif rich == "deep":
        case, desc = random_deep_case(seed)
    else:
        case, desc = random_case(seed, grids=(rich == "grids"), many=(rich == "many"))     # also "domains"
    if clipbox:
        # a clip box on top of the second family (world space; with a voxel-space transform the world is the unit cube at -p/ext)
        rng = np.random.default_rng(0xC11B000 + seed)
        if case.xfm is not None:
            wlo = -np.asarray(case.xfm["p"]) / np.array([case.xfm["vx"][0], case.xfm["vy"][1], case.xfm["vz"][2]])
            whi = wlo + 1.0
        else:
            wlo, whi = (np.asarray(v, dtype=np.float64) for v in case.scene.bounds())
        clo = wlo + rng.uniform(0.0, 0.45, 3) * (whi - wlo)
        case.clip = (list(clo), list(clo + rng.uniform(0.25, 0.6, 3) * (whi - wlo)))
        desc["clip"] = case.clip
    if rich == "domains":
        # the TF domain per channel: degenerate, reversed, very narrow, far wider than the data, off to one side
        rng = np.random.default_rng(0xD0A1000 + seed)
        case.xf_domains = [[(0.0, 0.0), (0.5, 0.5), (1.0, 0.0), (0.4, 0.4001), (-5.0, 5.0), (0.9, 3.0), (-2.0, 0.1), (0.0, 1e-30),
                            (0.25, 0.75)][int(rng.integers(0, 9))] for _ in case.scene.fields]
        desc["xf_domains"] = case.xf_domains
    if holes:
        # the eighth family: any of the above with cells missing (ids -1), on a scene marked allowEmptyCells — the reference's
        # ALLOW_EMPTY_CELLS build: poison value in the scalar buffers and the value ranges, corners skipped by the sampler
        from owlexabrick_amd import scenes
        rng = np.random.default_rng(0xE3F7000 + seed)
        frac = float(rng.choice([0.02, 0.1, 0.3, 0.6]))
        case.scene = scenes.with_empty_cells(case.scene, fraction=frac, seed=seed)
        case.allow_empty_cells, case.basis_form = True, 0
        desc["empty_fraction"] = frac
    case.fast_math = 0
    o = case.run_oracle(frames=frames)
    for accel in (1, 2, 0):              # kd-tree stack walk, kd-tree rope walk, LBVH restart
        case.accel = accel
        h = case.run_hip(stats=True, frames=frames)
        r = compare(o, h)
        if case.ao:      # AO directions go through cosf/sinf (libm vs OCML): a few rays may flip hit/miss
            da = np.abs(o[1] - h[1]).max(axis=-1)
            if (da > frames * ACCUM_ATOL).sum() > max(2, 0.003 * da.size * frames):
                bad.append(f"accel {accel} (AO): {r}")
            # the work must still be the same up to those few rays (a walk that loses segments shows here first)
            loose = [(k, o[2][k], h[2][k]) for k in STAT_KEYS if abs(o[2][k] - h[2][k]) > 0.01 * o[2][k] + 64]
            if loose:
                bad.append(f"accel {accel} (AO): counters {loose}")
            continue
        # a ray whose opacity crosses 0.98 within an ulp of powf may stop one sample earlier or later on one side
        # (tests/common.py: FLIP_BOUND, FLIP_FRACTION; 2 of 2000 seeds have such a pixel): then the pixel is bounded
        # by the flip tolerance and the counters by a few samples, otherwise everything is exact
        flipped = r["flip_pixels"] > 0 and r["flips_ok"] and r["rgba_bad"] <= 3 * r["flip_pixels"]
        if (r["accum_bad"] or r["rgba_bad"]) and not flipped:
            bad.append(f"accel {accel}: {r}")
        if flipped:
            slack = {k: 16 * r["flip_pixels"] * (8 if k == "corner_loads" else 1) for k in STAT_KEYS}
            if any(abs(o[2][k] - h[2][k]) > slack[k] for k in STAT_KEYS):
                bad.append(f"accel {accel}: counters beyond a flipped pixel {[(k, o[2][k], h[2][k]) for k in STAT_KEYS if o[2][k] != h[2][k]]}")
        elif {k: o[2][k] for k in STAT_KEYS} != {k: h[2][k] for k in STAT_KEYS}:
            # the same termination flip can leave the pixel inside the tolerance (the extra samples were transparent):
            # one ray's worth of work at most, and only in the large scenes
            slack = {k: (16 if k == "corner_loads" else 2) * 4 for k in STAT_KEYS}
            if o[2]["samples"] < 1e5 or any(abs(o[2][k] - h[2][k]) > slack[k] for k in STAT_KEYS):
                bad.append(f"accel {accel}: counters {[(k, o[2][k], h[2][k]) for k in STAT_KEYS if o[2][k] != h[2][k]]}")
        if h[2]["diag"][8] != 0:
            bad.append(f"accel {accel}: {h[2]['diag'][8]} slab-test mismatches")
        plain = case.run_hip(frames=frames)
        if not (np.array_equal(plain[1].view(np.uint32), h[1].view(np.uint32)) and np.array_equal(plain[0], h[0])):
            bad.append(f"accel {accel}: shipped kernel differs from the counting variant")
    case.accel, case.fast_math = None, None       # the defaults a caller gets (the module chooses the walk per frame)
    h = case.run_hip(stats=True, frames=frames)
    r = compare(o, h)
    if case.ao:
        da = np.abs(o[1] - h[1]).max(axis=-1)
        if (da > FLIP_BOUND * frames).sum() > max(2, 0.003 * da.size * frames):
            bad.append(f"defaults (AO): {r}")
    elif not (r["flips_ok"] and r["rgba_bad"] <= 3 * r["flip_pixels"]):
        bad.append(f"defaults: {r}")
    return bad, desc


if __name__ == "__main__":
    first, last = int(sys.argv[1]), int(sys.argv[2])
    keep = "--keep-going" in sys.argv
    rich = True if "--rich" in sys.argv else ("grids" if "--grids" in sys.argv else ("many" if "--many" in sys.argv else ("deep" if "--deep" in sys.argv else ("domains" if "--domains" in sys.argv else False))))
    fails, t0 = 0, time.time()
    for seed in range(first, last + 1):
        bad, desc = check(seed, rich, clipbox=("--clip" in sys.argv), holes=("--holes" in sys.argv))
        if bad:
            fails += 1
            print(f"FAIL seed {seed}: {desc}", flush=True)
            for b in bad:
                print("    ", b, flush=True)
            if not keep:
                break
        elif seed % 10 == 0:
            print(f"seed {seed} ok ({time.time() - t0:.0f}s)", flush=True)
    print(f"{last - first + 1 - fails if keep or not fails else '?'} passed, {fails} failed, {time.time() - t0:.0f}s", flush=True)
    sys.exit(1 if fails else 0)
