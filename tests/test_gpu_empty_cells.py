"""Scenes with empty cells (the reference's ALLOW_EMPTY_CELLS build, see tests/test_empty_cells.py) on the GPU: the
source-order kernels with the poison test (exa_kernels_f0e.o) against the oracle, both walks; shipped == counting variant;
the module refuses the per-axis form for such a scene."""
import numpy as np
import pytest

from common import Case, band_xf, compare
from owlexabrick_amd import scenes

pytestmark = pytest.mark.gpu

STAT_KEYS = ["segments", "sample_evals", "samples", "brick_visits", "corner_loads", "iso_segments", "iso_evals"]


def _amr():
    return scenes.amr(seed=3, root=(3, 3, 2), B=4, levels=3)


def _holes(sc, frac=0.15, seed=1):
    return scenes.with_empty_cells(sc, fraction=frac, seed=seed)


CASES = {
    "ex3_grad": lambda: Case(_holes(scenes.example("ex3")), W=96, H=64, grad=1, allow_empty_cells=True),
    "ex4_iso": lambda: Case(_holes(scenes.example("ex4"), 0.1), W=96, H=64, grad=1, iso=[(0.4, 0)], allow_empty_cells=True),
    "c1_64": lambda: Case(_holes(scenes.example("c1_64"), 0.3), W=128, H=128, allow_empty_cells=True),
    "amr": lambda: Case(_holes(_amr()), W=128, H=128, allow_empty_cells=True),
    "amr_grad_band": lambda: Case(_holes(_amr(), 0.4, 2), W=128, H=128, grad=1, xf=band_xf(), allow_empty_cells=True),
    "amr_iso_ragged": lambda: Case(_holes(_amr(), 0.05, 3), W=83, H=61, grad=1, iso=[(0.45, 0)], allow_empty_cells=True),
    "amr_2ch": lambda: Case(_holes(scenes.amr(seed=5, root=(2, 2, 2), B=4, levels=3, feature="plume", fields=2)), W=96, H=96, grad=1,
                            allow_empty_cells=True),
    "amr_contour": lambda: Case(_holes(_amr()), W=96, H=96, grad=1, opacity_scale=0.05, contour=[([1, 0.3, 0.2], 0.45, 0)],
                                allow_empty_cells=True),
    "gen_exajet": lambda: Case(_holes(scenes.generated(kind="exajet", seed=11, root=(4, 2, 2), B=8, levels=3), 0.2), W=160, H=96, grad=1,
                               allow_empty_cells=True),
}


@pytest.mark.parametrize("accel", [1, 0], ids=["kd", "lbvh"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_matches_oracle_with_empty_cells(name, accel):
    case = CASES[name]()
    case.accel, case.fast_math = accel, 0
    o = case.run_oracle()
    h = case.run_hip(stats=True)
    r = compare(o, h, name)
    assert r["accum_bad"] == 0 and r["rgba_bad"] == 0, r
    assert {k: o[2][k] for k in STAT_KEYS} == {k: h[2][k] for k in STAT_KEYS}
    assert np.isfinite(h[1]).all() and np.abs(h[1]).max() < 10.0          # no poison value in a pixel
    assert o[1][..., :3].sum() > 0


@pytest.mark.parametrize("fast_math", [0, 1])
@pytest.mark.parametrize("name", sorted(CASES))
def test_shipped_kernel_equals_counting_variant_with_empty_cells(name, fast_math):
    case = CASES[name]()
    case.fast_math = fast_math
    plain, counted = case.run_hip(frames=2), case.run_hip(frames=2, stats=True)
    assert np.array_equal(plain[1].view(np.uint32), counted[1].view(np.uint32))
    assert np.array_equal(plain[0], counted[0])


def test_holes_change_the_picture_and_the_option_is_needed():
    full = Case(_amr(), W=96, H=96, grad=1).run_hip()
    holes = Case(_holes(_amr(), 0.4, 2), W=96, H=96, grad=1, allow_empty_cells=True).run_hip()
    assert np.abs(full[1] - holes[1]).max() > 1e-2
    with pytest.raises(RuntimeError, match="overflow in index vector"):
        Case(_holes(_amr()), W=32, H=32).hip_renderer()


def test_per_axis_form_is_refused_for_a_scene_with_empty_cells():
    case = Case(_holes(scenes.example("ex3")), W=32, H=32, allow_empty_cells=True)
    R = case.hip_renderer()
    with pytest.raises(RuntimeError, match="keeps basis_form 0"):
        R.setOption("basis_form", 1)
    R.setOption("basis_form", 0)
    R.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_case_with_empty_cells_matches_oracle(seed):
    """the eighth seeded family (tests/gpu_fuzz.py --holes): the first three families' scenes with 2..60 % of the cells missing"""
    from gpu_fuzz import check
    bad, desc = check(seed, rich=(False, True, "grids")[seed % 3], holes=True)
    assert not bad, (desc, bad)
