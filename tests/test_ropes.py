"""The leaves and neighbour links of the rope walk (owlexabrick_amd/csrc/exa_ropes.h), checked on the CPU through the
diagnostic entry point exa_prep_ropes — the arrays the module uploads for option walk = 2:

* every region leaf's box is its domain, float for float; gaps (empty child slots of the kd-tree) are leaves of their own;
* every link leads to something that lies directly across the face, covers the whole face, and cannot be pushed further
  down without ambiguity; a link is "outside" exactly on the faces of the root box;
* a walk along the links — the kernel's rule: leave through the face with the smallest exit distance, descend behind an
  inner-node link by "far child when the plane's distance is <= the distance at which the previous leaf was left" —
  visits, for random rays, exactly the leaves a brute-force slab test over ALL leaves finds, in the order of their entry
  distances (the reference's closest-hit search, programs/exabrick.cu:184-238, 1675-1699, restated without any tree)."""
import numpy as np
import pytest

from owlexabrick_amd import binding, scenes

OUTSIDE = binding.KD_EMPTY + 1


def _scenes():
    yield "ex3", scenes.example("ex3")
    yield "ex4", scenes.example("ex4")
    yield "c1_64", scenes.example("c1_64")                       # one region: the root is a leaf
    yield "amr", scenes.amr(seed=3, root=(3, 3, 2), B=4, levels=3)
    yield "amr_2ch", scenes.amr(seed=5, root=(2, 2, 2), B=4, levels=3, feature="plume", fields=2)
    yield "gen_exajet", scenes.generated(kind="exajet", seed=11, root=(4, 2, 2), B=8, levels=3)
    from fuzz_cases import random_case
    for seed in (1, 7, 23):                                      # random brick partitions with holes: gaps in the tree
        case, _ = random_case(seed, grids=True)
        yield f"grids{seed}", case.scene


def _node_boxes(P, r):
    """box of every inner node of the tree behind the links, from the root box down"""
    nodes = r["nodes"]
    reg = P.regions()
    lo = np.minimum.reduce(reg["dom_lo"]).astype(np.float32) if len(reg) else None
    hi = np.maximum.reduce(reg["dom_hi"]).astype(np.float32)
    boxes = {}
    root = int(P.scene.kdRoot)
    stack = [(root, lo.copy(), hi.copy())]
    while stack:
        ref, l, h = stack.pop()
        if ref < 0:
            continue
        boxes[ref] = (l, h)
        n = nodes[ref]
        a = int(n["axis"])
        hl, lr = h.copy(), l.copy()
        hl[a] = n["split"]
        lr[a] = n["split"]
        stack.append((int(n["left"]), l, hl))
        stack.append((int(n["right"]), lr, h))
    return (lo, hi), boxes


@pytest.mark.parametrize("name,sc", list(_scenes()), ids=[n for n, _ in _scenes()])
def test_leaf_boxes_and_links(name, sc):
    P = binding.Prep(sc)
    r = P.ropes()
    reg = P.regions()
    nr = len(reg)
    assert r["flags"] == 3                                        # boxes == domains, planes on the short division's grid
    assert np.array_equal(r["boxes"][:nr, :3], reg["dom_lo"]) and np.array_equal(r["boxes"][:nr, 3:], reg["dom_hi"])
    assert np.array_equal(r["region"][:nr], np.arange(nr)) and (r["region"][nr:] == -1).all()
    assert len(r["nodes"]) == P.scene.numKdNodes
    kd = P.kd_nodes()
    gaps = int((kd["left"] == binding.KD_EMPTY).sum() + (kd["right"] == binding.KD_EMPTY).sum())
    assert len(r["boxes"]) == nr + gaps
    assert not (r["nodes"]["left"] == binding.KD_EMPTY).any() and not (r["nodes"]["right"] == binding.KD_EMPTY).any()
    (rlo, rhi), nbox = _node_boxes(P, r)
    nodes = r["nodes"]
    for i in range(len(r["boxes"])):
        lo, hi = r["boxes"][i, :3], r["boxes"][i, 3:]
        assert (lo < hi).all()
        for f in range(6):
            a, upper = f >> 1, f & 1
            t = int(r["links"][i, f])
            on_root = (hi[a] == rhi[a]) if upper else (lo[a] == rlo[a])
            assert (t == OUTSIDE) == bool(on_root), (i, f, t)
            if t == OUTSIDE:
                continue
            tlo, thi = (r["boxes"][~t, :3], r["boxes"][~t, 3:]) if t < 0 else nbox[t]
            # directly across the face ...
            assert (tlo[a] == hi[a]) if upper else (thi[a] == lo[a]), (i, f, t)
            # ... and over the whole face
            for b in range(3):
                if b != a:
                    assert tlo[b] <= lo[b] and thi[b] >= hi[b], (i, f, t, b)
            if t >= 0:
                # an inner node stays a link only when its plane cuts the face
                n = nodes[t]
                b = int(n["axis"])
                assert b != a and lo[b] < n["split"] < hi[b], (i, f, t)
    P.close()


def _rope_walk(r, root, rootbox, o, d, tmin=0.0):
    """the kernel's walk (exa_kernels.hip: ropeStep) in double precision: list of (leaf, t0, t1) with t0 < t1"""
    boxes, links, nodes = r["boxes"].astype(np.float64), r["links"], r["nodes"]
    lo, hi = rootbox
    with np.errstate(divide="ignore", invalid="ignore"):
        l, h = (lo - o) / d, (hi - o) / d
    r0, r1 = np.fmax.reduce(np.fmin(l, h)), np.fmin.reduce(np.fmax(l, h))
    tn = max(r0, tmin)
    if not tn < r1:
        return []
    out, ref, guard = [], root, 0
    while ref != OUTSIDE:
        guard += 1
        assert guard < 100000
        while ref >= 0:                                           # descend behind a link
            n = nodes[ref]
            a = int(n["axis"])
            if d[a] == 0.0:
                go_right = not (o[a] < n["split"])
            else:
                ts = (float(n["split"]) - o[a]) / d[a]
                go_right = (ts <= tn) == (d[a] > 0.0)
            ref = int(n["right"] if go_right else n["left"])
        leaf = ~ref
        with np.errstate(divide="ignore", invalid="ignore"):
            l, h = (boxes[leaf, :3] - o) / d, (boxes[leaf, 3:] - o) / d
        near, far = np.fmin(l, h), np.fmax(l, h)
        t_out = np.fmin.reduce(far)
        t0, t1 = max(tmin, np.fmax.reduce(near)), min(r1, t_out)
        if t0 < t1:
            out.append((leaf, t0, t1))
        if not t_out < r1:
            break
        a = int(np.argmax(far == t_out))
        ref = int(links[leaf, 2 * a + (1 if d[a] > 0.0 else 0)])
        tn = t_out
    return out


@pytest.mark.parametrize("name,sc", list(_scenes()), ids=[n for n, _ in _scenes()])
def test_walk_along_the_links_finds_what_a_slab_test_over_all_leaves_finds(name, sc):
    P = binding.Prep(sc)
    r = P.ropes()
    reg = P.regions()
    rlo = np.minimum.reduce(reg["dom_lo"]).astype(np.float64)
    rhi = np.maximum.reduce(reg["dom_hi"]).astype(np.float64)
    boxes = r["boxes"].astype(np.float64)
    rng = np.random.default_rng(0x120FE + len(boxes))
    centre, span = 0.5 * (rlo + rhi), rhi - rlo
    for k in range(60):
        if k % 3 == 0:                                            # from outside, towards a point inside
            o = centre + rng.uniform(1.0, 3.0) * span * rng.choice([-1.0, 1.0], 3) * rng.uniform(0.3, 1.0, 3)
            tgt = rlo + rng.uniform(0.05, 0.95, 3) * span
        elif k % 3 == 1:                                          # from inside the grid
            o = rlo + rng.uniform(0.05, 0.95, 3) * span
            tgt = rlo + rng.uniform(-0.5, 1.5, 3) * span
        else:                                                     # axis-parallel in one or two components (d = 0 there)
            o = rlo + rng.uniform(0.05, 0.95, 3) * span + 0.013
            tgt = o.copy()
            ax = rng.permutation(3)[: int(rng.integers(1, 3))]
            tgt[ax] += rng.choice([-1.0, 1.0], len(ax)) * span[ax]
        d = tgt - o
        d /= np.linalg.norm(d)
        got = _rope_walk(r, int(P.scene.kdRoot), (rlo, rhi), o, d)
        # brute force: the slab interval of EVERY leaf, those with t0 < t1, by entry distance
        with np.errstate(divide="ignore", invalid="ignore"):
            l, h = (boxes[:, :3] - o) / d, (boxes[:, 3:] - o) / d
        t0 = np.maximum(0.0, np.fmax.reduce(np.fmin(l, h), axis=1))
        t1 = np.fmin.reduce(np.fmax(l, h), axis=1)
        hit = np.nonzero(t0 < t1)[0]
        want = sorted(((int(i), float(t0[i]), float(t1[i])) for i in hit), key=lambda e: (e[1], e[2]))
        # slivers of rounding-error length are a matter of which side of an ulp a plane falls on in double precision
        eps = 1e-9 * max(1.0, float(np.abs(o).max()), float(span.max()))
        got_s = [e for e in got if e[2] - e[1] > eps]
        want_s = [e for e in want if e[2] - e[1] > eps]
        assert [e[0] for e in got_s] == [e[0] for e in want_s], (name, k, o.tolist(), d.tolist())
        assert np.allclose([e[1:] for e in got_s], [e[1:] for e in want_s], rtol=0, atol=0) or not got_s
    P.close()
