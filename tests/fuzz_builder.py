"""Seeded random inputs for exaBuilder (CPU only):  python tests/fuzz_builder.py FIRST LAST
Random partitions of a box into dense single-level blocks (tests/fuzz_cases._random_grids: any shape, levels 0..2, holes),
expanded into cells and shuffled, sometimes moved to negative coordinates, sometimes with exact duplicates; the three
builder types and a random --max-leaf-width.  The C++ tool must produce, byte for byte, what the restatement of
builder/builder.cpp produces (or fail with the same message), and the file must be a valid ExaBricks input: every cell in
exactly one brick, bricks dense and single-level."""
import sys
import time

import numpy as np

from fuzz_cases import _random_grids
from oracle import builder_oracle as bo
from test_builder import FLAGS, run_builder


def random_cells(seed):
    rng = np.random.default_rng(0xB01D000 + seed)
    grids, _ = _random_grids(rng)
    out = []
    for mx, my, mz, nx, ny, nz, lvl, *_ in grids:
        k = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"), -1).reshape(-1, 3)
        out.append(np.concatenate([np.array([mx, my, mz]) + k * (1 << lvl), np.full((len(k), 1), lvl)], axis=1))
    cells = np.concatenate(out).astype(np.int32)
    if rng.uniform() < 0.3:
        cells = cells + np.array([-4 * int(rng.integers(0, 9)), -4 * int(rng.integers(0, 9)), -4 * int(rng.integers(0, 9)), 0], dtype=np.int32)
    cells = cells[rng.permutation(len(cells))]
    if rng.uniform() < 0.2:
        cells = np.concatenate([cells, cells[rng.integers(0, len(cells), int(rng.integers(1, 6)))]])
    btype = [bo.SAH_ALIKE, bo.SPATIAL_MEDIAN, bo.SMALL_BRICK_COUNT][int(rng.integers(0, 3))]
    width = int(rng.choice([127, 127, 2, 3, 5, 8]))
    return cells, btype, width


def check(seed):
    cells, btype, width = random_cells(seed)
    r, data = run_builder(cells, FLAGS[btype], max_leaf_width=(None if width == 127 else width))
    try:
        exp = bo.to_bricks_file_bytes(bo.build_bricks(cells, btype, max_leaf_width=width))
    except RuntimeError as e:
        return ([] if (r.returncode == 1 and str(e) in r.stderr) else [f"restatement raises '{e}', tool rc {r.returncode}: {r.stderr[-200:]}"]), (len(cells), btype, width)
    bad = []
    if r.returncode != 0:
        bad.append(f"tool failed: {r.stderr[-300:]}")
    elif data != exp:
        bad.append("bytes differ from the restatement")
    else:
        a = np.frombuffer(data, dtype=np.int32)
        uniq = np.unique(cells, axis=0)
        index = {tuple(c): i for i, c in enumerate(cells.tolist())}
        seen, at = set(), 0
        while at < len(a):
            sx, sy, sz, x, y, z, l = (int(v) for v in a[at:at + 7])
            n = sx * sy * sz
            ids = a[at + 7:at + 7 + n].reshape(sz, sy, sx)
            at += 7 + n
            if max(sx, sy, sz) > width:
                bad.append("brick wider than --max-leaf-width")
            for k in range(sz):
                for j in range(sy):
                    for i in range(sx):
                        c = tuple(int(v) for v in cells[ids[k, j, i]])
                        if c != (x + (i << l), y + (j << l), z + (k << l), l):
                            bad.append(f"cell id {ids[k, j, i]} is not the cell at its place in the brick")
                        seen.add(c)
        if len(seen) != len(uniq):
            bad.append(f"{len(uniq)} distinct cells in, {len(seen)} in bricks")
    return bad[:3], (len(cells), btype, width)


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
