"""The ORACLE against the definitions on seeded random brick partitions (CPU only):  python tests/fuzz_oracle.py FIRST LAST [--per-axis]
Random partitions into bricks of any shape and level with holes (tests/fuzz_cases._random_grids).  Three checks that do
not share code with the oracle's march: (a) samplePoint[WithDerivative] == the hat-basis reconstruction summed over ALL
cells of the overlapping bricks (test_oracle_kat._hat_reconstruction); (b) the regions are a disjoint partition of the
union of the basis domains and each lists exactly the bricks whose domain contains it, with finestLevelCellWidth = the
finest of them (Regions.cpp:73-179, 293-299); (c) the oracle's pruned region search equals a brute-force slab test."""
import sys
import time

import numpy as np

from common import po
from fuzz_cases import _random_grids
from owlexabrick_amd import scenes
from test_oracle_kat import _hat_reconstruction


def check(seed, basis_form=0):
    rng = np.random.default_rng(0x04AC1E00 + seed)
    grids, ext = _random_grids(rng)
    sc = scenes.artificial(grids, name=f"grids{seed}")
    S = po.OracleScene(sc.bricks7, sc.cellIDs, sc.fields)
    S.set_basis_form(basis_form)
    R, LL = S.regions(), S.leaflist()
    bad = []
    # (a) hat basis
    picks = rng.choice(len(R), size=min(25, len(R)), replace=False)
    for r in picks:
        lo, hi = np.asarray(R[r]["dom_lo"], dtype=np.float64), np.asarray(R[r]["dom_hi"], dtype=np.float64)
        for p in rng.uniform(lo + 1e-3 * (hi - lo), hi - 1e-3 * (hi - lo), size=(3, 3)):
            ok, v, g = S.sample_point(int(r), p.astype(np.float32), with_derivative=True)
            eok, ev, eg = _hat_reconstruction(S, int(r), p.astype(np.float32))
            if ok != eok:
                bad.append(f"region {r}: valid {ok} vs {eok}")
            elif ok and (abs(float(v) - ev) > 2e-5 * max(1.0, abs(ev))
                         or not np.allclose(g, eg, rtol=2e-3, atol=2e-4 * max(1.0, np.abs(eg).max()))):
                bad.append(f"region {r} at {p}: {float(v)} {g} vs {ev} {eg}")
    # (b) partition
    b = sc.bricks7.astype(np.float64)
    cw = 2.0 ** b[:, 6]
    dlo = b[:, 3:6] - 0.5 * cw[:, None]
    dhi = b[:, 3:6] + (b[:, 0:3] + 0.5) * cw[:, None]
    for p in rng.uniform(dlo.min(axis=0), dhi.max(axis=0), size=(400, 3)):
        bricks = set(np.nonzero(((p > dlo) & (p < dhi)).all(axis=1))[0].tolist())
        inside = np.nonzero(((p > R["dom_lo"]) & (p < R["dom_hi"])).all(axis=1))[0]
        if not bricks:
            if len(inside):
                bad.append(f"{p}: region {inside} where no basis domain is")
            continue
        if len(inside) == 0:
            continue                                   # exactly on a face
        if len(inside) != 1:
            bad.append(f"{p}: in {len(inside)} regions")
            continue
        r = R[inside[0]]
        got = LL[r["leafListBegin"]:r["leafListBegin"] + r["leafListSize"]].tolist()
        if set(got) != bricks or got != sorted(set(got)):
            bad.append(f"{p}: region lists {got}, domains containing it {sorted(bricks)}")
        if r["finestLevelCellWidth"] != 2.0 ** sc.bricks7[got, 6].min():
            bad.append(f"{p}: finestLevelCellWidth {r['finestLevelCellWidth']}")
    # (c) region search
    lo, hi = S.voxel_bounds()
    active = (rng.uniform(size=S.num_regions) < 0.6).astype(np.uint8)
    for _ in range(150):
        o = rng.uniform(np.asarray(lo) - 6, np.asarray(hi) + 6)
        d = rng.normal(size=3)
        if rng.uniform() < 0.25:
            d[rng.integers(3)] = 0.0
        if not np.any(d):
            continue
        d /= np.linalg.norm(d)
        r, t0, t1 = S.trace_region(active, o, d, 1e-6, 1e8)
        if r == -2:
            bad.append(f"pruned region search differs from brute force for o={o} d={d}")
    return bad[:4], dict(seed=seed, bricks=len(grids), regions=len(R))


if __name__ == "__main__":
    first, last = int(sys.argv[1]), int(sys.argv[2])
    form = 1 if "--per-axis" in sys.argv else 0          # association of the basis sums (or_set_basis_form)
    fails, t0 = 0, time.time()
    for seed in range(first, last + 1):
        bad, desc = check(seed, basis_form=form)
        if bad:
            fails += 1
            print(f"FAIL seed {seed}: {desc} {bad}", flush=True)
    print(f"{fails} failed of {last - first + 1}, {time.time() - t0:.0f}s")
    sys.exit(1 if fails else 0)
