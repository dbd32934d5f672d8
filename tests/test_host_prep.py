"""Host logic behind the C ABI (no GPU): exa_prep_* against the oracle, bit for bit;
error behaviour; the library exports every symbol include/exa_hip.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from common import ROOT, po
from owlexabrick_amd import binding, scenes


def all_scenes():
    yield from (scenes.example(n) for n in ("ex0", "ex1", "ex2", "ex3", "ex4"))
    yield scenes.amr(seed=3, root=(3, 3, 2), B=4, levels=3)
    yield scenes.amr(seed=5, root=(2, 2, 2), B=4, levels=3, feature="plume", fields=2)
    yield scenes.amr(seed=8, root=(2, 3, 2), B=2, levels=4)            # tiny bricks, 4 levels
    yield scenes.generated(kind="exajet", seed=11, root=(4, 2, 2), B=8, levels=3)
    yield scenes.generated(kind="gear", seed=12, root=(2, 3, 2), B=8, levels=3, fields=2)


@pytest.mark.parametrize("sc", list(all_scenes()), ids=lambda s: s.name)
def test_prep_equals_oracle_bit_for_bit(sc):
    S = po.OracleScene(sc.bricks7, sc.cellIDs, sc.fields)
    for nt in (1, 5):
        P = binding.Prep(sc, num_threads=nt)
        assert P.bricks().tobytes() == S.bricks().tobytes()            # flatten (a5)
        assert P.scalars().tobytes() == S.scalars().tobytes()          # gather (a5)
        assert P.leaflist().tobytes() == S.leaflist().tobytes()        # buildFrom order (a3)
        assert P.regions().tobytes() == S.regions().tobytes()          # domains, ranges, flcw (a3, a4)
        lo, hi = P.voxel_bounds()
        olo, ohi = S.voxel_bounds()
        assert np.array_equal(lo, olo) and np.array_equal(hi, ohi)
        P.close()


def test_single_field_region_ranges():
    sc = scenes.amr(seed=5, root=(2, 2, 2), B=4, levels=3, feature="plume", fields=2)
    S = po.OracleScene(sc.bricks7, sc.cellIDs, sc.fields, num_region_fields=1)
    P = binding.Prep(sc, num_region_fields=1)
    assert P.regions().tobytes() == S.regions().tobytes()
    S2 = po.OracleScene(sc.bricks7, sc.cellIDs, sc.fields)             # all fields merged (Regions.cpp:190)
    assert (S2.regions()["vr_hi"] >= S.regions()["vr_hi"]).all()


def test_prep_errors_mirror_reference():
    sc = scenes.example("ex3")
    bad = scenes.Scene(sc.bricks7, sc.cellIDs[:-1], sc.fields)
    with pytest.raises(RuntimeError, match="sanity-check in brick size"):
        binding.Prep(bad)
    ids = sc.cellIDs.copy()
    ids[3] = 10 ** 6
    with pytest.raises(RuntimeError, match="invalid cell ID"):
        binding.Prep(scenes.Scene(sc.bricks7, ids, sc.fields))
    ids[3] = -1
    with pytest.raises(RuntimeError, match="overflow in index vector"):
        binding.Prep(scenes.Scene(sc.bricks7, ids, sc.fields))
    # the oracle reports the same conditions
    with pytest.raises(RuntimeError, match="sanity-check in brick size"):
        po.OracleScene(bad.bricks7, bad.cellIDs, bad.fields)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "exa_hip.h")).read()
    declared = set(re.findall(r"\b(exa_(?:hip|prep)_[a-z_]+)\s*\(", hdr))
    assert declared == set(binding.ABI_SYMBOLS)
    L = C.CDLL(binding.LIB_PATH)
    for sym in declared:
        assert hasattr(L, sym), sym


def test_abi_struct_sizes_match_header():
    assert C.sizeof(binding.ExaHipFrameState) == C.sizeof(po.FrameState)
    assert binding.BRICK_DTYPE.itemsize == 32 and binding.REGION_DTYPE.itemsize == 44
    assert C.sizeof(binding.ExaHipParams) == 28


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    P = binding.Prep(scenes.example("ex0"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        binding.Renderer(P)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "owlexabrick_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in src and "exa_oracle" not in src and "libexa_oracle" not in src, f


def test_artificial_restatement_matches_reference_tool_output():
    # fixtures = outputs of the reference's own exaArtificial on its own ex*.grids
    g = os.path.join(ROOT, "tests", "golden", "ref_artificial")
    for name in ("ex0", "ex1", "ex2", "ex3", "ex4"):
        cells = np.fromfile(os.path.join(g, name + ".cells"), dtype=np.int32).reshape(-1, 4)
        scal = np.fromfile(os.path.join(g, name + ".scalars"), dtype=np.float32)
        sc = scenes.example(name)
        assert np.array_equal(sc.meta["cells"], cells)
        assert np.array_equal(sc.fields[0], scal)


@pytest.mark.parametrize("sc", list(all_scenes()), ids=lambda s: s.name)
def test_exported_kd_tree_reproduces_region_domains(sc):
    """the recursion tree of buildRec: walking it from the union box, cutting at every
    split plane, must land on each region's domain exactly, every region exactly once."""
    P = binding.Prep(sc, num_threads=3)
    R, K, root = P.regions(), P.kd_nodes(), P.scene.kdRoot
    lo = R["dom_lo"].min(axis=0).astype(np.float32)
    hi = R["dom_hi"].max(axis=0).astype(np.float32)
    seen = np.zeros(len(R), dtype=np.int32)
    stack = [(root, lo.copy(), hi.copy())]
    while stack:
        ref, l, h = stack.pop()
        if ref == binding.KD_EMPTY:
            continue
        if ref < 0:
            r = ~ref
            seen[r] += 1
            assert np.array_equal(R[r]["dom_lo"], l) and np.array_equal(R[r]["dom_hi"], h)
            continue
        n = K[ref]
        assert l[n["axis"]] < n["split"] < h[n["axis"]]
        assert (n["left"] < 0 or n["left"] > ref) and (n["right"] < 0 or n["right"] > ref)   # preorder
        hl, lr = h.copy(), l.copy()
        hl[n["axis"]] = n["split"]
        lr[n["axis"]] = n["split"]
        stack.append((int(n["left"]), l, hl))
        stack.append((int(n["right"]), lr, h))
    assert (seen == 1).all()


@pytest.mark.parametrize("seed", range(30))
def test_prep_on_seeded_random_scenes(seed):
    """tests/fuzz_prep.py: the scenes of the random case families, exa_prep_* against the oracle byte for byte"""
    from fuzz_prep import check
    bad, desc = check(seed)
    assert not bad, (desc, bad)
