"""Seeded random scenes through the host preparation (CPU only): python tests/fuzz_prep.py FIRST LAST
The scenes of the three case families of tests/fuzz_cases.py; exa_prep_* (parallel region build, value ranges, scalar
gather, 1 and 5 threads) must equal the oracle's serial restatement of Regions::buildFrom / computeValueRange byte for
byte, and the kd tree handed to the module must be the same front-to-back partition (checked through the leaf count)."""
import sys
import time

import numpy as np

from common import po
from fuzz_cases import random_case, random_rich_case
from owlexabrick_amd import binding


def check(seed):
    fam = seed % 3
    if fam == 0:
        case, desc = random_case(seed)
    elif fam == 1:
        case, desc = random_case(seed, grids=True)
    else:
        case, desc, _ = random_rich_case(seed)
    sc = case.scene
    nrf = len(sc.fields) if case.multi else 1
    S = po.OracleScene(sc.bricks7, sc.cellIDs, sc.fields, num_region_fields=nrf)
    bad = []
    for nt in (1, 5):
        P = binding.Prep(sc, num_threads=nt, num_region_fields=nrf)
        for what, a, b in (("bricks", P.bricks(), S.bricks()), ("scalars", P.scalars(), S.scalars()),
                           ("leaf list", P.leaflist(), S.leaflist()), ("regions", P.regions(), S.regions())):
            if a.tobytes() != b.tobytes():
                bad.append(f"{what} differ ({nt} threads)")
        lo, hi = P.voxel_bounds()
        olo, ohi = S.voxel_bounds()
        if not (np.array_equal(lo, olo) and np.array_equal(hi, ohi)):
            bad.append("voxel bounds differ")
        P.close()
    return bad, desc


if __name__ == "__main__":
    first, last = int(sys.argv[1]), int(sys.argv[2])
    fails, t0 = 0, time.time()
    for seed in range(first, last + 1):
        bad, desc = check(seed)
        if bad:
            fails += 1
            print(f"FAIL seed {seed}: {desc} {bad}", flush=True)
    print(f"{fails} failed of {last - first + 1}, {time.time() - t0:.0f}s")
    sys.exit(1 if fails else 0)
