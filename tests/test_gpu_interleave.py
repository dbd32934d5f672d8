"""Where the cells lie in memory and how a frame is launched (round 3) — none of it may change a pixel:
* `interleave` (default on): the module re-lays the primary channels as float[cell][channel] on the device and the
  multi-channel DVR march evaluates all channels of a sample per brick visit.  Same values, same sums in the same order:
  the frame must equal the field-by-field march bit for bit, and the oracle within the stated tolerance (the seeded
  families of tests/test_gpu_fuzz.py cover the latter with 2..4 fields);
* `brick_order`: the cells re-laid along a Morton curve of the brick centres;
* `prepass_split`: tiles with long iso marches in their own pre-pass + march pipeline;
* `ao_defer`: the AO rays of the shaded hits traced by their own launch over a compact hit list."""
import numpy as np
import pytest

from common import Case, band_xf, compare
from owlexabrick_amd import scenes

pytestmark = pytest.mark.gpu


def _scene(nf, seed=5):
    return scenes.amr(seed=seed, root=(3, 2, 2), B=4, levels=3, fields=nf)


def _run(case, **options):
    case.options = options
    return case.run_hip(frames=2)


@pytest.mark.parametrize("nf", [2, 3, 4])
@pytest.mark.parametrize("grad", [0, 1])
@pytest.mark.parametrize("fast_math", [0, 1])
def test_interleaved_march_equals_field_by_field_march(nf, grad, fast_math):
    kw = dict(W=120, H=88, grad=grad, fast_math=fast_math, xf=[band_xf(0.2 + 0.1 * c, 0.9) for c in range(nf)], opacity_scale=0.6)
    ref = _run(Case(_scene(nf), **kw), interleave=0)
    for opts in (dict(interleave=1), dict(interleave=1, addr64=1)):
        got = _run(Case(_scene(nf), **kw), **opts)
        assert np.array_equal(ref[0], got[0]), opts
        assert np.array_equal(ref[1].view(np.uint32), got[1].view(np.uint32)), opts


@pytest.mark.parametrize("nf", [2, 3])
def test_interleaved_march_with_surfaces_clip_and_shards(nf):
    """iso-surface pre-pass in front (it samples the field-major arrays), a clip box, and a sharded handle"""
    from owlexabrick_amd import binding
    sc = _scene(nf, seed=9)
    kw = dict(W=96, H=96, grad=1, iso=[(0.45, nf - 1)], clip=None)
    ref = _run(Case(sc, **kw), interleave=0)
    got = _run(Case(sc, **kw), interleave=1)
    assert np.array_equal(ref[1].view(np.uint32), got[1].view(np.uint32))
    o = Case(sc, **kw, fast_math=0).run_oracle(frames=2)
    case = Case(sc, **kw, fast_math=0)
    case.options = dict(interleave=1)
    h = case.run_hip(frames=2)
    r = compare(o, h)
    assert r["flips_ok"] and r["rgba_bad"] <= 3 * r["flip_pixels"] and (r["accum_bad"] == 0 or r["flip_pixels"] > 0), r


def test_changing_the_number_of_primary_channels_rebuilds_the_copy():
    """multiFieldDvr on -> off -> on (numPrimaryChannels 3 -> 1 -> 3) on one handle equals fresh handles"""
    sc = _scene(3, seed=2)
    case = Case(sc, W=64, H=64, grad=1)
    R = case.hip_renderer()
    a3 = R.render().copy()
    R.params.numPrimaryChannels = 1
    a1 = R.render().copy()
    R.params.numPrimaryChannels = 2
    a2 = R.render().copy()
    R.params.numPrimaryChannels = 3
    assert np.array_equal(R.render(), a3)
    R.close()
    for n, want in ((1, a1), (2, a2)):
        c2 = Case(sc, W=64, H=64, grad=1)
        c2.options = dict(interleave=0)
        R2 = c2.hip_renderer()
        R2.params.numPrimaryChannels = n
        assert np.array_equal(R2.render(), want), n
        R2.close()


@pytest.mark.parametrize("nf", [1, 3])
def test_brick_order_in_memory_does_not_change_the_frame(nf):
    """option brick_order: the cells re-laid along a Morton curve of the brick centres (and back) on one handle; a scene
    whose brick list is shuffled, so that the two orders really differ"""
    sc = _scene(nf, seed=4)
    rng = np.random.default_rng(7)
    nb = sc.bricks7.shape[0]
    perm = rng.permutation(nb)
    vol = sc.bricks7[:, 0] * sc.bricks7[:, 1] * sc.bricks7[:, 2]
    begin = np.concatenate([[0], np.cumsum(vol)])[:-1]
    ids = np.concatenate([np.asarray(sc.cellIDs[begin[b]:begin[b] + vol[b]]) for b in perm])
    sc.bricks7 = np.ascontiguousarray(sc.bricks7[perm])
    sc.cellIDs = np.ascontiguousarray(ids)
    kw = dict(W=96, H=72, grad=1, iso=[(0.4, 0)])
    case = Case(sc, **kw)
    R = case.hip_renderer()
    ref = R.render().copy()
    for order in (1, 0, 1):
        R.setOption("brick_order", order)
        assert np.array_equal(R.render(), ref), order
    rgba, st = R.renderStats()                       # the counting variant and the literal sampler read the brick records
    assert np.array_equal(rgba, ref)
    R.setOption("accel", 0)                          # LBVH path: literal addBasisFunctions on the brick records
    lb = R.render().copy()
    R.setOption("brick_order", 0)
    assert np.array_equal(R.render(), lb)
    R.close()


@pytest.mark.parametrize("size", [(320, 256), (312, 250)], ids=["whole_tiles", "ragged_edge_tiles"])
def test_prepass_split_launch_plan_does_not_change_the_frame(size):
    """option prepass_split: heavy pre-pass tiles in their own pre-pass + march pipeline beside the rest of the frame
    (ragged: W and H not multiples of the 16-pixel tile — the heavy pipeline's deferred-AO list then starts behind the
    padded pixels of the cheap pipeline's edge tiles)"""
    sc = scenes.config("c3_gear", scale=0.2)
    kw = dict(W=size[0], H=size[1], grad=1, iso=[(0.5, 0)], ao=1, ao_length=200.0, xf_domains=[(0.0, 1.0)] * len(sc.fields))
    outs = {}
    for split in (0, 1, 2, 3):                   # 2: the split plan with the AO rays traced inline, 3: deferred and sorted
        case = Case(sc, **kw)
        case.options = dict(prepass_split=min(split, 1), ao_defer={0: 0, 1: 1, 2: 0, 3: 2}[split])
        R = case.hip_renderer()
        frames = []
        for f in range(4):                       # frame 0 measures the costs, the later ones run the plan
            R.updateFrameID(f)
            frames.append(R.render().copy())
        outs[split] = (frames, R.readAccum().copy())
        R.close()
    for k in (1, 2, 3):
        for f in range(4):
            assert np.array_equal(outs[0][0][f], outs[k][0][f]), (k, f)
        assert np.array_equal(outs[0][1].view(np.uint32), outs[k][1].view(np.uint32)), k


@pytest.mark.parametrize("world", [1, 3])
def test_deferred_ao_rays_equal_inline_ao_rays(world):
    """option ao_defer: the AO rays of the shaded hits traced by their own launch over a compact hit list — with a mesh in
    the scene (generic pre-pass), ragged tiles, several accumulated frames and a sharded handle"""
    sc = _scene(2, seed=11)
    lo, hi = sc.bounds()
    c = 0.5 * (lo + hi)
    tri = (np.array([[c[0] - 9, c[1] - 7, c[2]], [c[0] + 11, c[1] - 5, c[2] + 3], [c[0], c[1] + 12, c[2] - 2]], dtype=np.float32),
           np.array([[0, 1, 2]], dtype=np.int32))
    for meshes in (None, [tri]):
        kw = dict(W=104, H=72, grad=1, iso=[(0.42, 0), (0.5, 1)], ao=1, ao_length=30.0, meshes=meshes)
        outs = []
        for defer in (0, 1, 2):
            case = Case(sc, **kw)
            case.options = dict(ao_defer=defer)
            R = case.hip_renderer()
            R.setShard(world - 1, world)
            for f in range(3):
                R.updateFrameID(f)
                img = R.render().copy()
            outs.append((img, R.readAccum().copy()))
            R.close()
        for k in (1, 2):                                 # 2: the listed rays sorted by pixel block and direction class first
            assert np.array_equal(outs[0][0], outs[k][0]), k
            assert np.array_equal(outs[0][1].view(np.uint32), outs[k][1].view(np.uint32)), k
