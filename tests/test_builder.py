"""exaBuilder (cells -> bricks): the C++ tool against the Python restatement of
builder/builder.cpp, byte for byte, plus structural properties of the output."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from common import ROOT
from oracle import builder_oracle as bo
from owlexabrick_amd import binding, scenes

EXE = os.path.join(ROOT, "owlexabrick_amd", "host", "exaBuilder")
FLAGS = {bo.SAH_ALIKE: [], bo.SPATIAL_MEDIAN: ["--spatial-median"], bo.SMALL_BRICK_COUNT: ["--large-bricks"]}


def run_builder(cells, flags=(), max_leaf_width=None):
    with tempfile.TemporaryDirectory() as d:
        inp, out = os.path.join(d, "in.cells"), os.path.join(d, "out.bricks")
        np.asarray(cells, dtype=np.int32).tofile(inp)
        cmd = [EXE, inp, "-o", out] + list(flags)
        if max_leaf_width:
            cmd += ["--max-leaf-width", str(max_leaf_width)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
        return r, (open(out, "rb").read() if os.path.exists(out) else b"")


def cells_of(scene):
    """expand every brick of a scene into its cells (x,y,z,level), shuffled"""
    out = []
    for sx, sy, sz, x, y, z, l in scene.bricks7:
        k = np.stack(np.meshgrid(np.arange(sx), np.arange(sy), np.arange(sz), indexing="ij"), -1).reshape(-1, 3)
        c = np.concatenate([np.array([x, y, z]) + k * (1 << l), np.full((len(k), 1), l)], axis=1)
        out.append(c)
    cells = np.concatenate(out).astype(np.int32)
    rng = np.random.default_rng(4)
    return cells[rng.permutation(len(cells))]


def cell_sets():
    g = os.path.join(ROOT, "tests", "golden", "ref_artificial")
    for n in ("ex0", "ex1", "ex2", "ex3", "ex4"):           # outputs of the reference's own exaArtificial
        yield n, np.fromfile(os.path.join(g, n + ".cells"), dtype=np.int32).reshape(-1, 4)
    yield "amr", cells_of(scenes.amr(seed=3, root=(2, 2, 1), B=4, levels=3))
    yield "amr_b2", cells_of(scenes.amr(seed=8, root=(2, 3, 2), B=2, levels=3))
    c = cells_of(scenes.amr(seed=5, root=(2, 1, 1), B=4, levels=2))
    yield "dups", np.concatenate([c, c[:7], c[100:103]])                 # exact repeats are dropped
    coarse = np.array([[0, 0, 0, 1], [8, 0, 0, 2]], dtype=np.int32)      # inner nodes over finer cells: only removed
    yield "inner_nodes", np.concatenate([c, coarse])                      # when adjacent in sort order (else: throws)
    yield "negative", cells_of(scenes.amr(seed=3, root=(2, 2, 1), B=4, levels=2)) + np.array([-13, -5, -9, 0], dtype=np.int32) * np.array([4, 4, 4, 0])


@pytest.mark.parametrize("btype", [bo.SAH_ALIKE, bo.SPATIAL_MEDIAN, bo.SMALL_BRICK_COUNT])
@pytest.mark.parametrize("name,cells", list(cell_sets()), ids=[n for n, _ in cell_sets()])
def test_builder_matches_restatement_byte_for_byte(name, cells, btype):
    r, data = run_builder(cells, FLAGS[btype])
    try:
        exp = bo.to_bricks_file_bytes(bo.build_bricks(cells, btype))
    except RuntimeError as e:          # the reference throws on such input; the tool must fail the same way
        assert r.returncode == 1 and str(e) in r.stderr
        return
    assert r.returncode == 0, r.stderr
    assert data == exp


def test_builder_output_is_a_valid_exabricks_input():
    cells = cells_of(scenes.amr(seed=3, root=(2, 2, 1), B=4, levels=3))
    r, data = run_builder(cells, max_leaf_width=6)
    assert r.returncode == 0, r.stderr
    a = np.frombuffer(data, dtype=np.int32)
    seen = np.zeros(len(cells), dtype=int)
    at, nb = 0, 0
    while at < len(a):
        sx, sy, sz, x, y, z, l = a[at:at + 7]
        n = sx * sy * sz
        ids = a[at + 7:at + 7 + n]
        at += 7 + n
        nb += 1
        assert max(sx, sy, sz) <= 6
        seen[ids] += 1
        c = cells[ids].reshape(sz, sy, sx, 4)
        assert (c[..., 3] == l).all()                                   # single level
        kz, ky, kx = np.meshgrid(np.arange(sz), np.arange(sy), np.arange(sx), indexing="ij")
        assert (c[..., 0] == x + kx * (1 << l)).all() and (c[..., 1] == y + ky * (1 << l)).all() and (c[..., 2] == z + kz * (1 << l)).all()
    assert (seen == 1).all() and nb > 1                                 # every cell in exactly one brick
    assert f"created {nb} bricks" in r.stdout


def test_builder_cli_errors():
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 1 and "no input file specified" in r.stderr
    r = subprocess.run([EXE, "x.cells", "-o", "y", "--spatial-median", "--large-bricks"], capture_output=True, text=True)
    assert r.returncode == 1 and "you gotta decide" in r.stderr
    r = subprocess.run([EXE, "x.cells", "-o", "y", "--bogus"], capture_output=True, text=True)
    assert r.returncode == 1 and "un-recognized cmdline arg" in r.stderr


def test_built_bricks_feed_the_host_prep():
    from owlexabrick_amd import binding
    cells = np.fromfile(os.path.join(ROOT, "tests", "golden", "ref_artificial", "ex3.cells"), dtype=np.int32).reshape(-1, 4)
    scal = np.fromfile(os.path.join(ROOT, "tests", "golden", "ref_artificial", "ex3.scalars"), dtype=np.float32)
    r, data = run_builder(cells)
    a = np.frombuffer(data, dtype=np.int32)
    b7, ids, at = [], [], 0
    while at < len(a):
        n = int(a[at] * a[at + 1] * a[at + 2])
        b7.append(a[at:at + 7]); ids.append(a[at + 7:at + 7 + n]); at += 7 + n
    sc = scenes.Scene(np.array(b7), np.concatenate(ids), [scal])
    P = binding.Prep(sc)
    assert P.scene.totalCells == len(cells) and P.scene.numRegions >= len(b7)


@pytest.mark.parametrize("seed", range(20))
def test_builder_on_seeded_random_cell_sets(seed):
    """tests/fuzz_builder.py: random partitions into dense blocks (holes, negative coordinates, duplicates), the three
    builder types, random --max-leaf-width: bytes equal to the restatement and a valid ExaBricks file"""
    from fuzz_builder import check
    bad, desc = check(seed)
    assert not bad, (desc, bad)


def test_generated_scene_on_builder_made_bricks_holds_the_same_cells():
    """bench.py --bricks-file: the procedural scene re-instantiated on the bricks exaBuilder makes of its cells
    (scenes.config(..., bricks7=...)) — every cell of the original scene appears once, with the value the field functions
    give at its position"""
    sc = scenes.config("c4_exajet", scale=0.06, threads=4)
    cells = cells_of(sc)
    r, data = run_builder(cells)
    assert r.returncode == 0, r.stderr
    raw = np.frombuffer(data, dtype=np.int32)
    hdrs, at = [], 0
    while at < raw.size:
        h = raw[at:at + 7]
        hdrs.append(h.copy())
        at += 7 + int(h[0]) * int(h[1]) * int(h[2])
    b7 = np.stack(hdrs)
    assert len(b7) < len(sc.bricks7)                                    # the builder merges the 8^3 blocks
    sb = scenes.config("c4_exajet", scale=0.06, threads=4, bricks7=b7)
    assert sb.num_cells == sc.num_cells and np.array_equal(sb.bricks7, b7)

    def keyed(scene):
        key, val, at = [], [], 0
        for sx, sy, sz, x, y, z, l in np.asarray(scene.bricks7, dtype=np.int64):
            n = sx * sy * sz
            k = np.stack(np.meshgrid(np.arange(sz), np.arange(sy), np.arange(sx), indexing="ij"), -1).reshape(-1, 3)[:, ::-1]   # x fastest
            p = np.array([x, y, z]) + k * (1 << l)
            key.append((p[:, 0] << 42) | (p[:, 1] << 21) | p[:, 2] | (np.int64(l) << 60))
            val.append(np.asarray(scene.fields[0])[np.asarray(scene.cellIDs[at:at + n])])
            at += n
        key, val = np.concatenate(key), np.concatenate(val)
        order = np.argsort(key)
        return key[order], val[order]
    ka, va = keyed(sc)
    kb, vb = keyed(sb)
    assert np.array_equal(ka, kb) and np.array_equal(va, vb)


@pytest.mark.parametrize("btype", [bo.SAH_ALIKE, bo.SPATIAL_MEDIAN, bo.SMALL_BRICK_COUNT])
@pytest.mark.parametrize("seed", range(4))
def test_builder_allows_partially_filled_bricks_with_empty_cells(seed, btype):
    """--allow-empty-cells = the reference built with -DALLOW_EMPTY_CELLS=1 (builder/builder.cpp:473-495): a single-level
    set that fits a leaf becomes one even when cells are missing; the holes keep the id -1.  Byte for byte the restatement,
    every input cell in exactly one brick slot, and the result loads with the renderer's empty-cells option."""
    rng = np.random.default_rng(0xB11D + seed)
    full = cells_of(scenes.amr(seed=3 + seed, root=(2, 2, 1), B=4, levels=2))
    cells = full[rng.uniform(size=len(full)) > 0.25]                       # a quarter of the cells missing
    r, data = run_builder(cells, FLAGS[btype] + ["--allow-empty-cells"], max_leaf_width=6)
    assert r.returncode == 0, r.stderr
    assert data == bo.to_bricks_file_bytes(bo.build_bricks(cells, btype, max_leaf_width=6, allow_empty_cells=True))
    # parse the .bricks stream: ids are a permutation of the input cells plus -1 holes
    a = np.frombuffer(data, dtype=np.int32)
    at, ids, b7 = 0, [], []
    while at < len(a):
        sx, sy, sz = a[at:at + 3]
        b7.append(a[at:at + 7])
        ids.append(a[at + 7:at + 7 + sx * sy * sz])
        at += 7 + sx * sy * sz
    ids = np.concatenate(ids)
    assert (ids == -1).sum() > 0 and sorted(ids[ids >= 0].tolist()) == list(range(len(cells)))
    # without the option the same cells need far more (fully filled) bricks
    r2, data2 = run_builder(cells, FLAGS[btype], max_leaf_width=6)
    assert r2.returncode == 0 and len(np.frombuffer(data2, dtype=np.int32)) > 0
    n2 = 0
    a2, at = np.frombuffer(data2, dtype=np.int32), 0
    while at < len(a2):
        at += 7 + int(a2[at]) * int(a2[at + 1]) * int(a2[at + 2]); n2 += 1
    assert n2 > len(b7)
    # the host preparation takes the holes with its empty-cells flag
    sc = scenes.Scene(np.array(b7, dtype=np.int32), ids.astype(np.int32), [rng.uniform(size=len(cells)).astype(np.float32)])
    P = binding.Prep(sc, allow_empty_cells=True)
    assert (P.scalars() == np.float32(-1e20)).sum() == (ids == -1).sum()
    P.close()
