"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on
identical inputs; against the committed goldens; and size-independent properties
at the benchmark's frame size.  Tolerances are stated in tests/common.py."""
import os
import sys

import numpy as np
import pytest

from common import ACCUM_ATOL, FLIP_BOUND, Case, ROOT, band_xf, compare, po
from owlexabrick_amd import binding, harness, scenes

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden import GOLDEN_CASES, crop, golden_window, make_case  # noqa: E402

STAT_KEYS = ["segments", "sample_evals", "samples", "brick_visits", "corner_loads", "iso_segments", "iso_evals"]


def _amr():
    return scenes.amr(seed=3, root=(3, 3, 2), B=4, levels=3)


def _meshes():
    """an octahedron inside the _amr() volume and an axis-aligned quad (degenerate box on one axis)"""
    c, r = np.array([24.0, 22.0, 14.0]), 9.0
    v = [c + r * np.array(d) for d in ([1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1])]
    t = [[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]]
    quad_v = [[5, 5, 6], [40, 5, 6], [40, 40, 6], [5, 40, 6]]
    return [(np.array(v), np.array(t)), (np.array(quad_v, dtype=np.float64), np.array([[0, 1, 2], [0, 2, 3]]))]


def _an_interior_x_plane(sc):
    """the x-plane of the region partition that most regions share as a lower face (a face of many boxes, strictly inside the grid)"""
    P = binding.Prep(sc)
    reg = P.regions()
    vals, counts = np.unique(reg["dom_lo"][:, 0], return_counts=True)
    inner = vals > reg["dom_lo"][:, 0].min()
    plane = float(vals[inner][np.argmax(counts[inner])])
    P.close()
    return plane


CASES = {
    "ex0": lambda: Case(scenes.example("ex0"), W=96, H=64),
    "ex1_grad": lambda: Case(scenes.example("ex1"), W=96, H=64, grad=1),
    "ex2": lambda: Case(scenes.example("ex2"), W=96, H=64),
    "ex3": lambda: Case(scenes.example("ex3"), W=96, H=64),
    "ex3_grad": lambda: Case(scenes.example("ex3"), W=96, H=64, grad=1),
    "ex4_grad": lambda: Case(scenes.example("ex4"), W=96, H=64, grad=1),
    "ex3_iso": lambda: Case(scenes.example("ex3"), W=96, H=64, grad=1, iso=[(0.4, 0)]),
    "ex4_iso_noshade": lambda: Case(scenes.example("ex4"), W=96, H=64, iso=[(0.5, 0)], grad_iso=0),
    "ex4_two_isos": lambda: Case(scenes.example("ex4"), W=96, H=64, grad=1, iso=[(0.3, 0), (0.7, 0)]),
    "c1_64": lambda: Case(scenes.example("c1_64"), W=128, H=128),
    "c1_64_grad_iso": lambda: Case(scenes.example("c1_64"), W=128, H=128, grad=1, iso=[(0.3, 0)]),
    "amr": lambda: Case(_amr(), W=128, H=128),
    "amr_grad": lambda: Case(_amr(), W=128, H=128, grad=1),
    "amr_band": lambda: Case(_amr(), W=128, H=128, xf=band_xf()),
    "amr_noskip": lambda: Case(_amr(), W=128, H=128, xf=band_xf(), space_skipping=0),
    "amr_iso": lambda: Case(_amr(), W=128, H=128, grad=1, iso=[(0.45, 0)]),
    "amr_clip": lambda: Case(_amr(), W=96, H=96, grad=1, clip=([10, 8, 4], [40, 40, 28])),
    "amr_dt025_scale": lambda: Case(_amr(), W=96, H=96, dt=0.25, opacity_scale=0.3),
    "amr_ragged": lambda: Case(_amr(), W=83, H=61, grad=1),                     # not a multiple of the tile
    "amr_2ch": lambda: Case(scenes.amr(seed=5, root=(2, 2, 2), B=4, levels=3, feature="plume", fields=2),
                            W=128, H=128, grad=1),
    "amr_2ch_colormap": lambda: Case(scenes.amr(seed=5, root=(2, 2, 2), B=4, levels=3, feature="plume", fields=2),
                                     W=96, H=96, grad=1, multi=False, iso=[(0.5, 0)]),
    # iso-surfaces on the second of two primary channels only / on both: the march skips the
    # iso sampling of channels nothing refers to (until a segment holds an opaque hit)
    "amr_2ch_iso_ch1": lambda: Case(scenes.amr(seed=5, root=(2, 2, 2), B=4, levels=3, feature="plume", fields=2),
                                    W=96, H=96, grad=1, iso=[(0.35, 1), (0.65, 1)]),
    "amr_2ch_iso_both": lambda: Case(scenes.amr(seed=5, root=(2, 2, 2), B=4, levels=3, feature="plume", fields=2),
                                     W=96, H=96, grad=0, iso=[(0.5, 0), (0.4, 1)]),
    "amr_xfm": lambda: Case(_amr(), W=96, H=96, grad=1,
                            xfm=dict(vx=[48, 0, 0], vy=[0, 48, 0], vz=[0, 0, 32], p=[0, 0, 0]),
                            camera=([0.2, 1.7, 2.0], [0.5, 0.5, 0.5], [0, 1, 0], 60.0)),
    "amr_xfm_iso": lambda: Case(_amr(), W=96, H=96, grad=1, iso=[(0.45, 0)],
                                xfm=dict(vx=[48, 0, 0], vy=[0, 48, 0], vz=[0, 0, 32], p=[0, 0, 0]),
                                camera=([0.2, 1.7, 2.0], [0.5, 0.5, 0.5], [0, 1, 0], 60.0)),
    "amr_xfm_iso_ao": lambda: Case(_amr(), W=64, H=64, grad=0, iso=[(0.45, 0)], ao=1, ao_length=0.2,
                                   xfm=dict(vx=[24, 0, 0], vy=[0, 24, 0], vz=[0, 0, 16], p=[1, 2, 3]),
                                   camera=([-0.3, 2.6, 3.4], [0.9, 0.9, 0.8], [0, 1, 0], 60.0)),
    "amr_contour": lambda: Case(_amr(), W=96, H=96, grad=1, opacity_scale=0.05,
                                contour=[([1, 0.3, 0.2], 0.45, 0), ([0, 1, 0], 0.3, 0)]),
    "amr_contour_iso": lambda: Case(_amr(), W=96, H=96, grad=0, opacity_scale=0.05, iso=[(0.45, 0)],
                                    contour=[([0.2, 0.1, 1], 0.5, 0), ([1, 0, 0], 0.7, 0), ([0, 1, 0.5], 0.4, 0)]),
    "ex3_contour": lambda: Case(scenes.example("ex3"), W=96, H=64, contour=[([1, 1, 0], 0.6, 0)]),
    "amr_mesh": lambda: Case(_amr(), W=96, H=96, grad=1, opacity_scale=0.05, meshes=_meshes()),
    "amr_mesh_iso_contour": lambda: Case(_amr(), W=96, H=96, grad=0, opacity_scale=0.05, meshes=_meshes(), iso=[(0.5, 0)],
                                         contour=[([1, 0, 0.3], 0.6, 0)]),
    "amr_mesh_ao": lambda: Case(_amr(), W=64, H=64, grad=1, opacity_scale=0.05, meshes=_meshes(), ao=1, ao_length=12.0),
    "amr_inside": lambda: Case(_amr(), W=96, H=96, grad=1, camera=([20.3, 22.1, 14.2], [30, 20, 10], [0, 1, 0], 80.0)),
    "gen_exajet": lambda: Case(scenes.generated(kind="exajet", seed=11, root=(4, 2, 2), B=8, levels=3), W=160, H=96, grad=1),
    # every ray parallel to the x-planes (direction.x exactly 0: the walks' d == 0 branches, the plain division of the rope walk),
    # from an origin between two planes, and from one that lies exactly IN a plane of the partition (x = 8: (plane - o) / d = 0 / 0,
    # which the reference's NaN-ignoring min / max turn into a miss of both neighbours, exabrick.cu:197-210)
    "amr_rays_parallel_to_x_planes": lambda: Case(_amr(), W=64, H=48, grad=1, camera=dict(
        pos=np.array([10.3, -20.0, 9.1], dtype=np.float32), dir00=np.array([0.0, 1.0, -0.12], dtype=np.float32),
        dirDu=np.array([0.0, 0.0, 0.004], dtype=np.float32), dirDv=np.array([0.0, 0.006, 0.0], dtype=np.float32))),
    "amr_rays_in_an_x_plane": lambda: Case(_amr(), W=64, H=48, grad=1, camera=dict(
        pos=np.array([_an_interior_x_plane(_amr()), -20.0, 9.1], dtype=np.float32), dir00=np.array([0.0, 1.0, -0.12], dtype=np.float32),
        dirDu=np.array([0.0, 0.0, 0.004], dtype=np.float32), dirDv=np.array([0.0, 0.006, 0.0], dtype=np.float32))),
    # ... and a camera far away with a tiny field of view: direction components below 2^-30 (outside the short division's range)
    "amr_far_camera_tiny_components": lambda: Case(_amr(), W=64, H=48, grad=1, camera=dict(
        pos=np.array([24.0, 20.0, -3.0e6], dtype=np.float32), dir00=np.array([-3.0e-10, -2.0e-10, 1.0], dtype=np.float32),
        dirDu=np.array([1.0e-11, 0.0, 0.0], dtype=np.float32), dirDv=np.array([0.0, 1.0e-11, 0.0], dtype=np.float32))),
}


@pytest.mark.parametrize("form", [0, 1], ids=["source_order", "per_axis"])
@pytest.mark.parametrize("accel", [1, 0, 2], ids=["kd", "lbvh", "rope"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_matches_oracle(name, accel, form):
    """form: the association of the eight-corner basis sums, the same on both sides: the kernels execute
    the oracle's operation sequence in either form, so tolerance and counters are the same"""
    case = CASES[name]()
    case.accel, case.basis_form = accel, form
    case.fast_math = 0          # library powf: the work counters must then match sample for sample
    o = case.run_oracle()
    h = case.run_hip(stats=True)
    r = compare(o, h, name)
    if case.ao:      # AO directions go through cosf/sinf (libm vs OCML): a few rays may flip hit/miss
        da = np.abs(o[1] - h[1]).max(axis=-1)
        assert (da > ACCUM_ATOL).sum() <= max(2, 0.003 * da.size), r
        return
    assert r["accum_bad"] == 0 and r["rgba_bad"] == 0, r
    assert {k: o[2][k] for k in STAT_KEYS} == {k: h[2][k] for k in STAT_KEYS}   # identical work, sample for sample
    assert h[2]["diag"][8] == 0        # kd interval == the reference's slab test, every leaf


@pytest.mark.parametrize("form", [0, 1], ids=["source_order", "per_axis"])
@pytest.mark.parametrize("fast_math", [0, 1])
@pytest.mark.parametrize("name", sorted(CASES))
def test_shipped_kernel_equals_instrumented_variant(name, fast_math, form):
    """the parity tests above run the counting variant of the kernels; the variant a caller gets from
    exa_hip_render must produce the same accumulation buffer bit for bit"""
    case = CASES[name]()
    case.fast_math, case.basis_form = fast_math, form
    plain, counted = case.run_hip(), case.run_hip(stats=True)
    assert np.array_equal(plain[1].view(np.uint32), counted[1].view(np.uint32))
    assert np.array_equal(plain[0], counted[0])


@pytest.mark.parametrize("form", [0, 1], ids=["source_order", "per_axis"])
@pytest.mark.parametrize("accel", [1, 0, 2], ids=["kd", "lbvh", "rope"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_shipped_defaults_within_stated_tolerance(name, accel, form):
    """every case of the matrix with the options a caller gets by default — fast_math=1 (hardware exp2/log2 opacity
    correction, 1-ulp rcp/sqrt in the sample epilogue of the kd march; the LBVH variant has no fast path) and
    tf_filter=1 — against the oracle, under the flip tolerance of tests/common.py"""
    case = CASES[name]()
    case.accel, case.basis_form = accel, form
    o, h = case.run_oracle(), case.run_hip(stats=True)
    r = compare(o, h, name)
    if case.ao:      # AO directions go through cosf/sinf (libm vs OCML): a few rays may flip hit/miss
        da = np.abs(o[1] - h[1]).max(axis=-1)
        assert (da > FLIP_BOUND).sum() <= max(2, 0.003 * da.size), r
        return
    assert r["flips_ok"] and r["rgba_bad"] <= 3 * r["flip_pixels"], r
    for k in ("samples", "brick_visits", "segments"):
        assert abs(o[2][k] - h[2][k]) <= 1e-3 * o[2][k] + 2, (k, o[2][k], h[2][k])


def _step_tf(at=64):
    xf = harness.default_xf()
    xf[:, 3] = (np.arange(128) >= at).astype(np.float32) * 0.5
    return xf


@pytest.mark.parametrize("accel", [1, 0], ids=["kd", "lbvh"])
def test_tf_filter_fixed_point_weight_matches_oracle_in_both_modes(accel):
    """tf_filter 1 (default) = the CUDA tex1D filter weight in 1.8 fixed point, 0 = full precision: a two-texel ramp in
    the opacity makes the weight itself visible; each mode equals the oracle in the same mode sample for sample, and
    the two modes differ from each other"""
    acc = {}
    for mode in (1, 0):
        case = Case(_amr(), W=96, H=96, grad=1, xf=_step_tf(40), tf_filter=mode, accel=accel, fast_math=0, opacity_scale=0.5)
        o, h = case.run_oracle(), case.run_hip(stats=True)
        r = compare(o, h, f"step tf, filter {mode}")
        assert r["accum_bad"] == 0 and r["rgba_bad"] == 0, r
        assert {k: o[2][k] for k in STAT_KEYS} == {k: h[2][k] for k in STAT_KEYS}
        acc[mode] = h[1]
    d = np.abs(acc[0] - acc[1])
    assert d.max() > 1e-4            # the quantised weight is visible ...
    assert d.max() < 0.05            # ... and small: at most 1/512 of a texel step per sample, plus rare termination flips
    # activity of the regions goes through the same lookup (activeForVolumeSampling, exabrick.cu:250-281)
    case = Case(_amr(), W=32, H=32, xf=_step_tf(40), accel=accel)
    S = case.oracle_scene()
    fs, P = case.oracle_state(S)
    R = case.hip_renderer()
    for mode in (1, 0):
        R.setOption("tf_filter", mode)
        S.set_tf_filter(mode)
        assert np.array_equal(R.readActivity(0), S.volume_active(fs, P))
    R.close()


@pytest.mark.parametrize("name", ["ex0", "ex2", "ex3_iso", "amr_grad", "amr_iso", "amr_2ch", "gen_exajet", "c4_small"])
def test_device_built_lbvh_is_the_host_built_tree(name):
    """accel=0 walks a software LBVH over the regions.  It is built on the device (Morton codes, radix sort, topology
    level by level; option lbvh_build=0, default); lbvh_build=1 builds it on the host.  Same codes, same split rule,
    same node numbering: identical frames, identical node fetches, and both equal the kd walk's frame."""
    if name == "c4_small":
        case = Case(scenes.config("c4_exajet", scale=0.2), W=512, H=512, grad=1, xf_domains=[(0.0, 1.0)])
    else:
        case = CASES[name]()
    case.accel, case.fast_math = 0, 0
    out = {}
    for where in (0, 1):
        case.options = {"lbvh_build": where}
        out[where] = case.run_hip(stats=True)
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))
    assert out[0][2]["nodes_visited"] == out[1][2]["nodes_visited"] > 0
    case.accel, case.options = 1, {}
    kd = case.run_hip()
    assert np.array_equal(kd[1].view(np.uint32), out[0][1].view(np.uint32))


def test_ao_rays_match_up_to_trig_ulps():
    # AO directions go through cosf/sinf (libm vs OCML differ by an ulp); a hit/miss flip moves a
    # pixel by 1/2 of its surface colour, so allow a handful of flipped pixels and nothing else
    case = Case(scenes.example("ex4"), W=96, H=64, iso=[(0.5, 0)], ao=1, ao_length=3.0)
    o, h = case.run_oracle(), case.run_hip()
    da = np.abs(o[1] - h[1]).max(axis=-1)
    assert (da > ACCUM_ATOL).sum() <= 0.002 * da.size
    assert o[1][..., :3].sum() > 0


@pytest.mark.parametrize("accel", [1, 2], ids=["kd", "rope"])
@pytest.mark.parametrize("name", ["amr_grad", "amr_2ch", "amr_iso", "ex4_grad", "amr_mesh_ao"])
def test_region_records_that_do_not_pack_change_no_pixel(name, accel):
    # option pack_records 0: the march takes region ids from the walk and loads the region records — what a scene does whose
    # records {first brick, brick count, level} do not fit a 32-bit leaf reference (BASELINE-size scenes all pack; the exajet-like
    # scene grown to 1.2e9 cells does not).  Same frame, same counters, with either walk.
    case, frames = CASES[name](), 2
    case.accel = accel
    ref = case.run_hip(frames=frames, stats=True)
    case.options = {"pack_records": 0}
    got = case.run_hip(frames=frames, stats=True)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1].view(np.uint32), ref[1].view(np.uint32))
    assert {k: got[2][k] for k in STAT_KEYS} == {k: ref[2][k] for k in STAT_KEYS}


@pytest.mark.parametrize("accel", [1, 2], ids=["kd", "rope"])
def test_ao_rays_beside_the_march_change_no_pixel(accel):
    # option ao_overlap: the AO launch in front of the march (0, the default) or on a side stream beside it, the pixels finished by
    # compositeKdKernel (1); deferred (ao_defer 1, default), sorted (2) or inline (0) rays: one frame, bit for bit, over 3 samples
    case = Case(_amr(), W=96, H=80, grad=1, iso=[(0.45, 0)], ao=1, ao_length=6.0)
    frames = {}
    for plan in ((0, 1), (1, 1), (1, 2), (0, 0)):
        # options are applied in order, after the walk the case's `accel` selects (tests/common.py)
        case.options = {"accel": 1, "walk": 1 if accel == 1 else 2, "ao_overlap": plan[0], "ao_defer": plan[1]}
        frames[plan] = case.run_hip(frames=3)
    ref = frames[(0, 1)]
    assert ref[1][..., :3].sum() > 0
    for plan, f in frames.items():
        assert np.array_equal(f[0], ref[0]) and np.array_equal(f[1].view(np.uint32), ref[1].view(np.uint32)), plan


@pytest.mark.parametrize("name", sorted(GOLDEN_CASES))
def test_hip_matches_golden_fixture(name):
    g = np.load(os.path.join(ROOT, "tests", "golden", f"oracle_{name}.npz"))
    case, frames = make_case(name)
    case.fast_math = 0
    rgba, acc, st = case.run_hip(frames=frames, stats=True)
    win = golden_window(name)                        # the C1 fixtures hold a crop of the 512x512 frame
    rgba, acc = np.ascontiguousarray(crop(rgba, win)), crop(acc, win)
    assert np.abs(acc - g["accum"]).max() <= ACCUM_ATOL
    d = np.abs(rgba.view(np.uint8).astype(int) - g["rgba"].view(np.uint8).astype(int))
    assert d.max() <= 1
    if win is None:                                  # the fixture's counters cover what the fixture holds
        gs = dict(zip([str(k) for k in g["stat_keys"]], g["stats"].tolist()))
        assert {k: gs[k] for k in STAT_KEYS} == {k: st[k] for k in STAT_KEYS}


def test_region_activity_matches_bounds_programs():
    case = Case(_amr(), W=32, H=32, xf=band_xf(), iso=[(0.45, 0)])
    S = case.oracle_scene()
    fs, P = case.oracle_state(S)
    R = case.hip_renderer()
    assert np.array_equal(R.readActivity(0), S.volume_active(fs, P))
    assert np.array_equal(R.readActivity(1), S.iso_active(fs))
    # updateXF re-evaluates activity (needVolumeBVHRebuild)
    xf = harness.default_xf()
    R.updateXF(0, xf[:, 3], xf[:, :3], case.xf_domains[0], 1.0)
    S.set_xf(0, xf)
    assert np.array_equal(R.readActivity(0), S.volume_active(fs, P))
    R.setSpaceSkipping(False)
    assert R.readActivity(0).all()
    R.close()


def test_state_changes_between_frames():
    # one renderer, a sequence of setter calls as the viewer would issue them
    case = Case(_amr(), W=64, H=64, grad=1, fast_math=0)
    R = case.hip_renderer()
    img0 = R.render()
    R.updateXF(0, band_xf()[:, 3], band_xf()[:, :3], case.xf_domains[0], 1.0)
    img1 = R.render()
    ref1 = Case(_amr(), W=64, H=64, grad=1, xf=band_xf()).run_oracle()
    assert compare(ref1, (img1, R.readAccum(), None))["accum_bad"] == 0
    R.updateIsoValues([0.45, 0], [0, 0], [1, 0])
    img2 = R.render()
    ref2 = Case(_amr(), W=64, H=64, grad=1, xf=band_xf(), iso=[(0.45, 0)]).run_oracle()
    assert compare(ref2, (img2, R.readAccum(), None))["accum_bad"] == 0
    R.updateIsoValues([0, 0], [0, 0], [0, 0])
    R.resizeFrameBuffer((48, 32))
    cam = harness.default_camera(*R.voxelSpaceBounds, 48, 32)
    R.updateCamera(cam["pos"], cam["dir00"], cam["dirDu"], cam["dirDv"])
    img3 = R.render()
    ref3 = Case(_amr(), W=48, H=32, grad=1, xf=band_xf()).run_oracle()
    assert compare(ref3, (img3, R.readAccum(), None))["accum_bad"] == 0
    assert not np.array_equal(img0[:32, :48], img3)
    R.close()


@pytest.mark.parametrize("what", ["1x1", "17x5", "more_ranks_than_tiles", "transparent_tf", "opaque_tf", "camera_inside"])
def test_scheduling_edge_cases(what):
    """launch-order feedback and wide march under degenerate frames: same pixels as the static one-lane launch
    over three frames (measure, reorder, steady state), nothing hangs or faults"""
    kw, shard = dict(W=96, H=64, grad=1), (0, 1)
    if what == "1x1":
        kw.update(W=1, H=1)
    elif what == "17x5":
        kw.update(W=17, H=5)
    elif what == "more_ranks_than_tiles":
        kw.update(W=40, H=24)               # 3 x 2 tiles
        shard = (7, 8)                      # this rank owns no tile at all
    elif what == "transparent_tf":
        xf = harness.default_xf(); xf[:, 3] = 0.0
        kw.update(xf=xf)
    elif what == "opaque_tf":
        xf = harness.default_xf(); xf[:, 3] = 1.0
        kw.update(xf=xf)
    elif what == "camera_inside":
        kw.update(camera=([20.3, 22.1, 14.2], [30, 20, 10], [0, 1, 0], 80.0))
    out = {}
    for sched in (0, 1):
        case = Case(_amr(), **kw)
        case.options = {"tile_feedback": sched, "wide_march": sched}
        R = case.hip_renderer()
        R.setShard(*shard)
        R.updateFrameID(0)
        frames = []
        for _ in range(3):
            rgba = R.render()
            frames.append((np.array(rgba, copy=True), R.readAccum().copy()))
        R.close()
        for f in frames[1:]:
            assert np.array_equal(f[0], frames[0][0]) and np.array_equal(f[1].view(np.uint32), frames[0][1].view(np.uint32))
        out[sched] = frames[0]
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))


@pytest.mark.parametrize("accel", [1, 0], ids=["kd", "lbvh"])
def test_clock_heat_map(accel):
    """frameState.clockScale > 0 (exabrick.cu:1703-1707): red = clockScale * cycles / 1e6 clamped to 1, green and
    blue untouched"""
    case = Case(_amr(), W=96, H=64, grad=1, accel=accel)
    plain = case.run_hip()[1]
    R = case.hip_renderer()
    R.frameState.clockScale = 1e9          # saturates: every pixel's program runs more than a thousandth of a cycle
    R.render()
    hot = R.readAccum()
    R.frameState.clockScale = 1e-3         # a million cycles would be 1e-3: tiny but positive for marched pixels
    R.render()
    warm = R.readAccum()
    R.close()
    assert np.array_equal(hot[..., 1:3], plain[..., 1:3]) and np.array_equal(warm[..., 1:3], plain[..., 1:3])
    assert np.all(hot[..., 0] == 1.0)
    assert np.all(warm[..., 0] > 0.0) and np.all(warm[..., 0] < 1.0)


def test_phase_time_variant_keeps_pixels_and_reports_cycles():
    """stats_mode 2: the shipped march plus a clock read at every phase change"""
    case = CASES["amr_grad"]()
    plain = case.run_hip()
    R = case.hip_renderer()
    R.setOption("stats_mode", 2)
    rgba, st = R.renderStats()
    acc = R.readAccum()
    R.close()
    assert np.array_equal(plain[0], rgba) and np.array_equal(plain[1].view(np.uint32), acc.view(np.uint32))
    pc = st["phase_cycles"]
    assert all(c > 0 for c in pc[:4]) and st["samples"] == 0      # times, no work counters


@pytest.mark.parametrize("lanes", [2, 4])
@pytest.mark.parametrize("name", sorted(CASES))
def test_wide_march_is_bit_identical(name, lanes):
    """wide_march = 2 / 4 marches every tile with that many lanes per ray (normally only the tiles on a
    frame's critical path): consecutive samples evaluated side by side, composited in order — the same
    accumulation buffer bit for bit, over 2 accumulated frames (scenes with several primary channels keep
    the one-lane march)"""
    out = {}
    for mode in (0, lanes):
        case = CASES[name]()
        case.options = {"wide_march": mode}
        out[mode] = case.run_hip(frames=2)
    assert np.array_equal(out[0][1].view(np.uint32), out[lanes][1].view(np.uint32))
    assert np.array_equal(out[0][0], out[lanes][0])


@pytest.mark.parametrize("world", [1, 3])
def test_launch_order_feedback_does_not_change_pixels(world):
    """the first frame after a state change records tile costs, later frames launch the heaviest tiles
    first (option tile_feedback): same pixels before and after, and with the feedback switched off"""
    case = Case(_amr(), W=200, H=136, grad=1)
    out = {}
    for fb in (1, 0):
        R = case.hip_renderer()
        R.setOption("tile_feedback", fb)
        R.setOption("wide_march", 1 if fb else 0)
        R.setShard(world - 1, world)
        R.updateFrameID(0)
        frames = [(R.render().copy(), R.readAccum().copy()) for _ in range(3)]
        for rgba, acc in frames[1:]:
            assert np.array_equal(rgba, frames[0][0]) and np.array_equal(acc.view(np.uint32), frames[0][1].view(np.uint32))
        out[fb] = frames[0]
        # a moving camera re-measures every frame (with whatever mix of one-lane and wide tiles is active)
        cams = [harness.camera([-30.0 - 4 * i, 70.0 + 3 * i, 90.0], [24, 22, 14], [0, 1, 0], 50.0, 200, 136) for i in range(3)]
        moved = []
        for cam in cams:
            R.updateCamera(cam["pos"], cam["dir00"], cam["dirDu"], cam["dirDv"])
            moved.append((R.render().copy(), R.readAccum().copy()))
        out[(fb, "moved")] = moved
        R.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))
    for a, b in zip(out[(0, "moved")], out[(1, "moved")]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


def test_progressive_accumulation_16_frames():
    case = Case(scenes.example("ex4"), W=64, H=48, grad=1, fast_math=0)
    o = case.run_oracle(frames=16)
    h = case.run_hip(frames=16)
    r = compare(o, h)
    assert np.abs(o[1] - h[1]).max() <= 16 * ACCUM_ATOL and r["rgba_bad"] == 0


def test_tile_shards_and_untile_reassemble_frame():
    import torch
    case = Case(_amr(), W=104, H=72, grad=1)          # ragged tiles
    full = case.run_hip()[0]
    for world in (2, 3, 8):
        stride = harness.shard_stride(104, 72, world)
        gathered = torch.zeros(stride * world, dtype=torch.int32, device="cuda")
        for rank in range(world):
            R = case.hip_renderer()
            R.setShard(rank, world)
            R.updateFrameID(0)
            shard = torch.zeros(stride, dtype=torch.int32, device="cuda")
            R.render(device_ptr=shard.data_ptr())
            gathered[rank * stride:(rank + 1) * stride] = shard
            if rank == 0:
                out = torch.zeros(104 * 72, dtype=torch.int32, device="cuda")
                root = R
            else:
                R.close()
        torch.cuda.synchronize()
        root.untile(gathered.data_ptr(), stride, world, out.data_ptr())
        torch.cuda.synchronize()
        img = out.cpu().numpy().view(np.uint32).reshape(72, 104)
        root.close()
        assert np.array_equal(img, full), world
        # the host mirror of the layout agrees with the device one
        assert np.array_equal(harness.untile(gathered.cpu().numpy().view(np.uint32), 104, 72, world), full)


def test_errors_are_reported_not_swallowed():
    case = Case(scenes.example("ex2"), W=32, H=32)
    R = case.hip_renderer()
    R.params.numPrimaryChannels = 3                     # more channels than scalar fields
    with pytest.raises(RuntimeError, match="channel counts"):
        R.render()
    R.params.numPrimaryChannels = 1
    R.params.dt = 0.0
    with pytest.raises(RuntimeError, match="dt must be"):
        R.render()
    R.params.dt = 0.5
    with pytest.raises(RuntimeError, match="mismatching xf size"):
        R.updateXF(0, np.zeros(10), np.zeros((10, 3)), (0, 1))
    with pytest.raises(RuntimeError, match="bad size"):
        R.resizeFrameBuffer((0, 5))
    R.close()


def test_region_cell_width_must_be_a_power_of_two():
    """the reference only writes finestLevelCellWidth = 1 << finestLevel (exa/Regions.cpp:293-299) and the march forms
    1/(dt*width) from the width's exponent bits: a hand-built scene with any other width is refused, not mis-sampled"""
    from owlexabrick_amd import binding
    prep = binding.Prep(scenes.example("ex3"))
    regs = prep.regions()
    keep = regs["finestLevelCellWidth"][0]
    for bad in (3.0, 0.5, 6.0, float("nan")):
        regs["finestLevelCellWidth"][0] = bad
        with pytest.raises(RuntimeError, match="power of two"):
            binding.Renderer(prep)
    regs["finestLevelCellWidth"][0] = keep
    binding.Renderer(prep).close()


def test_full_frame_properties_at_benchmark_size():
    """2048x2048 on an exajet-like scene: properties that do not need the oracle at full size,
    plus an oracle check on a crop."""
    sc = scenes.config("c4_exajet", scale=0.2)
    case = Case(sc, W=2048, H=2048, grad=1, xf_domains=[(0.0, 1.0)])
    R = case.hip_renderer()
    R.setOption("tile_order", 0)
    R.setOption("accel", 0)
    lb = R.render()
    R.setOption("accel", 1)
    R.setOption("fast_math", 0)
    a = R.render()
    assert np.array_equal(a, lb)                              # kd walk and LBVH restart pick the same segments
    R.setOption("fast_math", 1)
    a = R.render()
    assert np.abs(harness.unpack_rgba8(a).astype(int) - harness.unpack_rgba8(lb).astype(int)).max() <= 6
    for order in (1, 4, 5):
        R.setOption("tile_order", order)
        R.resizeFrameBuffer((2048, 2048))
        b = R.render()
        assert np.array_equal(a, b)                           # launch order never changes pixels
    acc_skip = R.readAccum()
    R.setSpaceSkipping(False)
    c = R.render()
    acc_noskip = R.readAccum()
    assert np.abs(acc_skip - acc_noskip).max() < 1e-5         # KAT-7 at full size
    assert np.abs(harness.unpack_rgba8(b).astype(int) - harness.unpack_rgba8(c).astype(int)).max() <= 1
    R.close()
    x0 = 1024 - 48
    o = case.run_oracle(window=(x0, x0, x0 + 96, x0 + 96))
    d = np.abs(o[1][x0:x0 + 96, x0:x0 + 96] - acc_skip[x0:x0 + 96, x0:x0 + 96])
    bad = (d > ACCUM_ATOL + 1e-4 * np.abs(o[1]).max()).any(axis=-1).sum()
    assert bad <= 5 and d.max() <= 0.021          # only termination flips, see tests/common.py
    assert o[2]["samples"] > 0


@pytest.mark.parametrize("accel", [1, 0], ids=["kd", "lbvh"])
def test_one_region_scene_skips_its_region_when_inactive(accel):
    """a scene of one region: the kd tree is a single leaf and no node carries the activity bits.  The region must still
    be skipped when the TF is transparent over its value range (no segment, as in the reference, whose volume BVH then
    holds no primitive) and the iso walk when no iso value lies in its range (found by the seeded random cases)"""
    sc = scenes.example("c1_64")                          # values in [0, 1]
    xf = harness.default_xf()
    xf[:, 3] = (np.arange(128) >= 120).astype(np.float32)  # opaque only above 0.94
    for kw, expect_seg in ((dict(xf=xf, xf_domains=[(0.0, 4.0)]), 0),            # the field maps to texels 0..32: transparent
                           (dict(xf=xf, xf_domains=[(0.0, 4.0)], space_skipping=0), None),
                           (dict(iso=[(2.5, 0)]), None)):                           # iso value outside the field's range
        case = Case(sc, W=64, H=48, accel=accel, fast_math=0, **kw)
        o, h = case.run_oracle(), case.run_hip(stats=True)
        assert compare(o, h)["accum_bad"] == 0
        assert {k: o[2][k] for k in STAT_KEYS} == {k: h[2][k] for k in STAT_KEYS}, kw
        if expect_seg is not None:
            assert h[2]["segments"] == expect_seg
        if "iso" in kw:
            assert h[2]["iso_segments"] == 0 and o[2]["iso_segments"] == 0


def test_walk_is_chosen_from_the_fraction_of_active_regions():
    """option walk = 0 (default): the rope walk when at least 40 % of the regions are active for the volume march, else the
    stack walk, which prunes inactive subtrees (the rope walk passes through every leaf on the ray).  The counting variant
    tells which one ran (walk_leaf_visits); either way the frame is the same bit for bit."""
    sc = _amr()
    dense, sparse = Case(sc, W=96, H=96, grad=1), Case(sc, W=96, H=96, grad=1, xf=band_xf(0.45, 0.5))
    for case, want_rope in ((dense, True), (sparse, False)):
        R = case.hip_renderer()
        frac = float(R.readActivity(0).mean())
        assert (frac >= 0.4) == want_rope, frac
        rgba, st = R.renderStats()
        acc = R.readAccum().copy()
        assert (st["walk_leaf_visits"] > 0) == want_rope and (st["walk_restarts"] == 0 or not want_rope)
        for walk in (1, 2):
            R.setOption("walk", walk)
            rgba_w, st_w = R.renderStats()
            assert (st_w["walk_leaf_visits"] > 0) == (walk == 2)
            assert np.array_equal(rgba_w, rgba) and np.array_equal(R.readAccum().view(np.uint32), acc.view(np.uint32))
            assert {k: st_w[k] for k in STAT_KEYS} == {k: st[k] for k in STAT_KEYS} and st_w["diag"][8] == 0
        R.close()
