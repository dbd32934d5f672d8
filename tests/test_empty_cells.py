"""The reference's build option ALLOW_EMPTY_CELLS (CMakeLists.txt:70-73, default OFF) as a property of the scene, CPU side:
cell id -1 = "no cell here" (exa/ExaBricks.cpp:46-49); its slot in the gathered scalar buffers holds
EMPTY_CELL_POISON_VALUE = -1e20f (exa/OptixRenderer.cpp:116-118, programs/FrameState.h:27), which the regions' value ranges
then include (computeValueRange reads every slot, exa/Regions.cpp:182-240), and addBasisFunctions skips such corners
(programs/exabrick.cu:614-618, 646 ...).  Oracle against the definition, host preparation against the oracle byte for byte,
error behaviour without the option.  (GPU == oracle: tests/test_gpu_empty_cells.py.)"""
import numpy as np
import pytest

from common import Case, po
from owlexabrick_amd import binding, scenes

POISON = np.float32(-1e20)


def _hat_reconstruction_with_holes(S, region, p):
    """sum over the NON-EMPTY cells c of the overlapping bricks of H_c(p) s_c / sum H_c(p): the basis of the ExaBrick
    paper with the empty cells' basis functions removed (what skipping a corner means)"""
    B, R, L, sc = S.bricks(), S.regions()[region], S.leaflist(), S.scalars()
    sw = swv = 0.0
    for b in L[R["leafListBegin"]:R["leafListBegin"] + R["leafListSize"]]:
        br = B[b]
        cw = float(1 << int(br["level"]))
        sx, sy, sz = (int(v) for v in br["size"])
        idx = np.arange(sx * sy * sz)
        ctr = np.stack([idx % sx, (idx // sx) % sy, idx // (sx * sy)], axis=1) * cw + np.asarray(br["lower"], dtype=np.float64) + 0.5 * cw
        h = np.maximum(0.0, 1.0 - np.abs((np.asarray(p, dtype=np.float64)[None] - ctr) / cw)).prod(axis=1)
        s = sc[int(br["begin"]) + idx]
        keep = s != POISON
        sw += h[keep].sum(); swv += (h[keep] * s[keep].astype(np.float64)).sum()
    return (False, 0.0) if sw <= 1e-20 else (True, swv / sw)


@pytest.mark.parametrize("scene", ["ex3", "ex4", "amr"])
def test_sample_point_skips_empty_cells(scene):
    base = scenes.example(scene) if scene != "amr" else scenes.amr(seed=3, root=(2, 2, 2), B=4, levels=3)
    sc = scenes.with_empty_cells(base, fraction=0.15, seed=1)
    S = po.OracleScene(sc.bricks7, sc.cellIDs, sc.fields, allow_empty_cells=True)
    assert (S.scalars() == POISON).sum() == (sc.cellIDs < 0).sum() > 0          # the gather writes the poison value
    R = S.regions()
    assert R["vr_lo"].min() == POISON                                            # ... and the value ranges see it
    rng = np.random.default_rng(3)
    checked = skipped = 0
    for r in rng.choice(len(R), size=min(60, len(R)), replace=False):
        lo, hi = np.asarray(R[r]["dom_lo"], dtype=np.float64), np.asarray(R[r]["dom_hi"], dtype=np.float64)
        for p in rng.uniform(lo + 1e-3 * (hi - lo), hi - 1e-3 * (hi - lo), size=(8 if len(R) < 60 else 4, 3)):
            ok, v, g = S.sample_point(int(r), p.astype(np.float32), with_derivative=True)
            eok, ev = _hat_reconstruction_with_holes(S, int(r), p.astype(np.float32))
            assert ok == eok, (r, p)
            if ok:
                assert abs(float(v) - ev) <= 2e-5 * max(1.0, abs(ev)) and abs(float(v)) < 1e3       # no poison leaks into a value
                assert np.all(np.isfinite(g)) and np.abs(g).max() < 1e6
                checked += 1
            else:
                skipped += 1
    assert checked > 100


def test_all_corners_empty_is_an_invalid_sample():
    sc = scenes.example("ex2")                      # one 8^3 brick
    ids = np.array(sc.cellIDs, copy=True)
    vol = ids.reshape(8, 8, 8)
    vol[2:4, 2:4, 2:4] = -1                          # the eight cells around (3,3,3)
    S = po.OracleScene(sc.bricks7, ids, sc.fields, allow_empty_cells=True)
    ok, _, _ = S.sample_point(0, np.array([3.0, 3.0, 3.0], dtype=np.float32))      # cell centres 2.5 / 3.5: all eight corners empty
    assert not ok
    ok, v, _ = S.sample_point(0, np.array([3.4, 3.0, 3.0], dtype=np.float32))      # still between the same cell centres
    assert not ok
    ok, v, _ = S.sample_point(0, np.array([3.9, 3.0, 3.0], dtype=np.float32))      # x now between 3.5 and 4.5: the high-x cells exist
    assert ok and abs(v) < 10


@pytest.mark.parametrize("nt", [1, 5])
def test_prep_equals_oracle_bit_for_bit_with_empty_cells(nt):
    for base in (scenes.example("ex4"), scenes.amr(seed=5, root=(2, 2, 2), B=4, levels=3, feature="plume", fields=2)):
        sc = scenes.with_empty_cells(base, fraction=0.2, seed=2)
        S = po.OracleScene(sc.bricks7, sc.cellIDs, sc.fields, allow_empty_cells=True)
        P = binding.Prep(sc, num_threads=nt, allow_empty_cells=True)
        assert P.scene.allowEmptyCells == 1
        assert P.scalars().tobytes() == S.scalars().tobytes()
        assert P.leaflist().tobytes() == S.leaflist().tobytes()
        assert P.regions().tobytes() == S.regions().tobytes()
        P.close()


def test_empty_cells_are_an_error_without_the_option():
    sc = scenes.with_empty_cells(scenes.example("ex3"), fraction=0.1, seed=0)
    with pytest.raises(RuntimeError, match="overflow in index vector"):          # exa/OptixRenderer.cpp:119-120
        binding.Prep(sc)
    with pytest.raises(RuntimeError, match="overflow in index vector"):
        po.OracleScene(sc.bricks7, sc.cellIDs, sc.fields)
    # with the option EVERY negative id is "no cell", as in the renderer (OptixRenderer.cpp:116-118; "-1 only" is an assert of
    # the loader, compiled out of a release build): prep and oracle agree byte for byte on such a file
    ids = np.array(sc.cellIDs, copy=True)
    ids[ids < 0] = -7
    odd = scenes.Scene(sc.bricks7, ids, sc.fields)
    P = binding.Prep(odd, allow_empty_cells=True)
    S = po.OracleScene(odd.bricks7, odd.cellIDs, odd.fields, allow_empty_cells=True)
    P1 = binding.Prep(sc, allow_empty_cells=True)
    assert P.scalars().tobytes() == S.scalars().tobytes() == P1.scalars().tobytes()
    assert P.regions().tobytes() == S.regions().tobytes()
    P.close(); P1.close()
    with pytest.raises(RuntimeError, match="overflow in index vector"):
        binding.Prep(odd)


def test_scene_with_empty_cells_keeps_the_source_order():
    sc = scenes.with_empty_cells(scenes.example("ex3"), fraction=0.1, seed=0)
    a = Case(sc, W=48, H=32, grad=1, allow_empty_cells=True).run_oracle()
    S = Case(sc, W=48, H=32, grad=1, allow_empty_cells=True).oracle_scene()
    S.set_basis_form(1)                                                          # refused silently: stays 0
    c = Case(sc, W=48, H=32, grad=1, allow_empty_cells=True)
    fs, P = c.oracle_state(S)
    _, acc, _ = S.render(fs, P, 48, 32)
    assert np.array_equal(acc, a[1])
    assert np.isfinite(a[1]).all() and a[1][..., :3].sum() > 0


def test_create_refuses_a_scene_struct_with_garbage_in_allow_empty_cells():
    """ExaHipScene.allowEmptyCells sits where the struct used to end in padding: a caller that fills the struct field by
    field without zeroing it passes whatever was there.  Anything but 0 / 1 is refused (before a device is touched),
    instead of silently switching the scene to the empty-cells kernels."""
    import ctypes as C
    L = binding.lib()
    P = binding.Prep(scenes.example("ex3"))
    sc = binding.ExaHipScene.from_buffer_copy(P.scene)
    sc.allowEmptyCells = 0x5a5a5a5a
    h = C.c_void_p()
    assert L.exa_hip_create(C.byref(sc), 0, C.byref(h)) != 0 and not h.value
    L.exa_hip_last_error.restype = C.c_char_p
    assert b"allowEmptyCells" in L.exa_hip_last_error(None)
    P.close()
