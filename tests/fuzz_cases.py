"""Seeded random parity cases: scene (block size, levels, feature, one or two fields), frame size, camera (outside,
inside, axis-aligned — rays parallel to the split planes), transfer function (ramp / band / random table / steps),
step size, opacity scale, gradient shading, iso-surfaces, clip box, space skipping, frame id.  `random_case(seed)` is a
pure function of the seed; tests/test_gpu_fuzz.py runs a fixed set under `-m gpu`, `python tests/gpu_fuzz.py A B` sweeps
any range of seeds on a GPU box."""
import numpy as np

from common import Case, band_xf
from owlexabrick_amd import harness, scenes


def _random_xf(rng, kind):
    xf = harness.default_xf()
    n = xf.shape[0]
    t = np.arange(n) / (n - 1.0)
    if kind == "band":
        lo = float(rng.uniform(0.05, 0.6))
        return band_xf(lo, lo + float(rng.uniform(0.1, 0.35)))
    if kind == "table":                                   # rough random table: every texel pair has a different slope
        xf[:, 3] = rng.uniform(0.0, 1.0, n).astype(np.float32) * (rng.uniform(0, 1, n) > 0.3)
        xf[:, :3] = rng.uniform(0.0, 1.0, (n, 3)).astype(np.float32)
    elif kind == "steps":                                 # piecewise constant with exact zeros (space skipping edges)
        k = int(rng.integers(3, 9))
        levels = rng.choice([0.0, 0.0, 0.15, 0.5, 1.0], size=k)
        xf[:, 3] = levels[np.minimum((t * k).astype(int), k - 1)].astype(np.float32)
    elif kind == "faint":
        xf[:, 3] = (0.02 * t).astype(np.float32)
    return xf


def _random_grids(rng):
    """a random partition of a box into bricks of any shape and level (the reference's `.grids` form, one brick per grid:
    tools/artificial): extents are multiples of 4 finest cells, a brick of level l has extent / 2^l cells per axis (so
    1-cell-thin and 127-free odd shapes occur), neighbours may differ by two levels, a brick is sometimes left out (a hole
    in the domain), values are trilinear between random corner values or constant."""
    ext = [int(rng.choice([8, 12, 16, 24])) for _ in range(3)]
    boxes = [((0, 0, 0), tuple(ext))]
    for _ in range(int(rng.integers(1, 10))):
        i = int(rng.integers(0, len(boxes)))
        lo, hi = boxes[i]
        ax = int(rng.integers(0, 3))
        n4 = (hi[ax] - lo[ax]) // 4
        if n4 < 2:
            continue
        cut = lo[ax] + 4 * int(rng.integers(1, n4))
        a_hi = list(hi); a_hi[ax] = cut
        b_lo = list(lo); b_lo[ax] = cut
        boxes[i] = (lo, tuple(a_hi))
        boxes.append((tuple(b_lo), hi))
    grids = []
    for lo, hi in boxes:
        if len(boxes) > 2 and rng.uniform() < 0.1:
            continue
        lvl = int(rng.integers(0, 3))
        dims = [(hi[k] - lo[k]) >> lvl for k in range(3)]
        vals = [float(rng.uniform(0, 1))] * 8 if rng.uniform() < 0.25 else [float(v) for v in rng.uniform(0, 1, 8)]
        grids.append([lo[0], lo[1], lo[2], dims[0], dims[1], dims[2], lvl] + vals)
    if not grids:                                         # every brick left out (one seed in a few thousand): keep one
        lo, hi = boxes[0]
        grids.append([lo[0], lo[1], lo[2], hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], 0] + [0.25] * 8)
    return grids, np.array(ext, dtype=np.float64)


def random_case(seed, grids=False, many=False):
    """grids=True: the third family — the same knobs on a random partition into bricks of any shape and level;
    many=True: three or four scalar fields (the march variant for more than two TF tables)"""
    rng = np.random.default_rng((0x6E1D5000 if grids else (0x3A4F000 if many else 0xE7A000)) + seed)
    B = int(rng.choice([2, 4, 4, 8]))
    levels = int(rng.integers(1, 4))
    root = tuple(int(v) for v in rng.integers(1, 4, 3))
    if B == 8:
        root = tuple(min(r, 2) for r in root)
    fields = int(rng.choice([1, 1, 1, 2]))
    if many:
        fields = int(rng.choice([3, 3, 4]))
    feature = str(rng.choice(["shell", "plume"]))
    if grids:
        g, ext = _random_grids(rng)
        scene = scenes.artificial(g, name=f"grids{seed}")
        fields, feature, root, B, levels = 1, "grids", tuple(int(e) for e in ext), 0, len(g)
    else:
        scene = scenes.amr(seed=int(rng.integers(1, 1 << 20)), root=root, B=B, levels=levels, feature=feature, fields=fields)
        ext = np.array(root, dtype=np.float64) * B * (1 << (levels - 1))
    W, H = int(rng.integers(17, 97)), int(rng.integers(9, 81))
    mode = str(rng.choice(["default", "outside", "inside", "axis", "grazing"]))
    fovy = float(rng.uniform(25.0, 95.0))
    centre = 0.5 * ext
    if mode == "default":
        camera = None
    elif mode == "outside":
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        camera = (list(centre + d * float(rng.uniform(0.9, 2.5)) * ext.max()), list(centre + rng.uniform(-0.2, 0.2, 3) * ext), [0, 1, 0], fovy)
    elif mode == "inside":
        camera = (list(rng.uniform(0.15, 0.85, 3) * ext), list(rng.uniform(0.0, 1.0, 3) * ext), [0, 1, 0], fovy)
    elif mode == "axis":                                   # centre ray exactly along an axis: zero direction components
        ax = int(rng.integers(0, 3)); sgn = float(rng.choice([-1.0, 1.0]))
        o = centre.copy(); o[ax] += sgn * 1.4 * ext[ax]
        o[(ax + 1) % 3] = float(np.round(o[(ax + 1) % 3]))  # on a cell face of the finest level
        at = o.copy(); at[ax] = centre[ax]
        camera = (list(o), list(at), [0, 0, 1] if ax == 1 else [0, 1, 0], fovy)
    else:                                                  # grazing: along a face of the volume
        o = np.array([-0.3 * ext[0], ext[1] * float(rng.choice([0.0, 1.0])), 0.5 * ext[2]])
        camera = (list(o), [ext[0], o[1], 0.5 * ext[2]], [0, 1, 0], fovy)
    xf_kind = str(rng.choice(["ramp", "band", "table", "steps", "faint"]))
    xfs = [_random_xf(rng, xf_kind) for _ in range(fields)]
    iso = None
    if rng.uniform() < 0.35:
        iso = [(float(rng.uniform(0.15, 0.85)), int(rng.integers(0, fields)))]
        if rng.uniform() < 0.3:
            iso.append((float(rng.uniform(0.15, 0.85)), int(rng.integers(0, fields))))
    clip = None
    if rng.uniform() < 0.2:
        lo = rng.uniform(0.0, 0.4, 3) * ext
        clip = (list(lo), list(lo + rng.uniform(0.3, 0.6, 3) * ext))
    kw = dict(W=W, H=H, grad=int(rng.integers(0, 2)), iso=iso, xf=xfs, dt=float(rng.choice([0.5, 0.5, 0.25, 1.0, 0.37, 2.0])),
              opacity_scale=float(rng.choice([1.0, 1.0, 0.3, 0.05])), space_skipping=int(rng.uniform() < 0.8), clip=clip,
              frameID=int(rng.choice([0, 0, 3])), camera=camera, grad_iso=int(rng.integers(0, 2)),
              multi=bool(fields == 1 or rng.uniform() < 0.7), xf_domains=[(0.0, 1.0)] * fields if rng.uniform() < 0.5 else None)
    desc = dict(seed=seed, B=B, levels=levels, root=root, fields=fields, feature=feature, camera=mode, xf=xf_kind,
                **{k: v for k, v in kw.items() if k not in ("xf", "camera", "xf_domains")})
    return Case(scene, **kw), desc


def _octahedron(centre, r):
    v = [np.asarray(centre) + r * np.array(d, dtype=np.float64) for d in ([1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1])]
    t = [[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]]
    return np.array(v), np.array(t)


def random_rich_case(seed):
    """the second family: the same knobs plus the surface features and the other scene generator — AO rays, contour
    planes, triangle meshes, a voxel-space transform with a world-space camera, the C++ scene generator's three kinds,
    several accumulated frames, the full-precision TF filter.  Returns (case, desc, frames)."""
    rng = np.random.default_rng(0x51C4000 + seed)
    fields = int(rng.choice([1, 1, 2]))
    if rng.uniform() < 0.3:
        kind = str(rng.choice(["lanl", "gear", "exajet"]))
        B = int(rng.choice([4, 8]))
        levels = int(rng.integers(2, 4))
        root = (int(rng.integers(1, 4)), int(rng.integers(1, 3)), int(rng.integers(1, 3)))
        scene = scenes.generated(kind=kind, seed=int(rng.integers(1, 1 << 20)), root=root, B=B, levels=levels, fields=fields)
        what = kind
    else:
        B = int(rng.choice([2, 4, 4]))
        levels = int(rng.integers(1, 4))
        root = tuple(int(v) for v in rng.integers(1, 4, 3))
        what = str(rng.choice(["shell", "plume"]))
        scene = scenes.amr(seed=int(rng.integers(1, 1 << 20)), root=root, B=B, levels=levels, feature=what, fields=fields)
    ext = np.array(root, dtype=np.float64) * B * (1 << (levels - 1))
    centre = 0.5 * ext
    W, H = int(rng.integers(24, 129)), int(rng.integers(16, 97))
    fovy = float(rng.uniform(30.0, 90.0))
    xfm = None
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    if rng.uniform() < 0.2:
        # voxel = vx*wx + vy*wy + vz*wz + p: the world is the unit cube moved by -p/ext; the camera is given in world space
        p = rng.uniform(-3.0, 3.0, 3)
        xfm = dict(vx=[float(ext[0]), 0, 0], vy=[0, float(ext[1]), 0], vz=[0, 0, float(ext[2])], p=[float(v) for v in p])
        wc = 0.5 - p / ext
        camera = (list(wc + d * float(rng.uniform(1.2, 2.5))), list(wc + rng.uniform(-0.1, 0.1, 3)), [0, 1, 0], fovy)
    elif rng.uniform() < 0.3:
        camera = (list(rng.uniform(0.15, 0.85, 3) * ext), list(rng.uniform(0.0, 1.0, 3) * ext), [0, 1, 0], fovy)
    else:
        camera = (list(centre + d * float(rng.uniform(0.9, 2.2)) * ext.max()), list(centre + rng.uniform(-0.2, 0.2, 3) * ext), [0, 1, 0], fovy)
    xf_kind = str(rng.choice(["ramp", "band", "table", "steps", "faint"]))
    xfs = [_random_xf(rng, xf_kind) for _ in range(fields)]
    iso = None
    if rng.uniform() < 0.5:
        iso = [(float(rng.uniform(0.15, 0.85)), int(rng.integers(0, fields)))]
        if rng.uniform() < 0.3:
            iso.append((float(rng.uniform(0.15, 0.85)), int(rng.integers(0, fields))))
    contour = None
    if rng.uniform() < 0.25:
        contour = []
        for _ in range(int(rng.integers(1, 4))):
            n = rng.normal(size=3)
            contour.append(([float(v) for v in n], float(rng.uniform(0.2, 0.8)), int(rng.integers(0, fields))))
    meshes = None
    if rng.uniform() < 0.2 and xfm is None:
        meshes = [_octahedron(centre + rng.uniform(-0.15, 0.15, 3) * ext, float(rng.uniform(0.1, 0.3)) * float(ext.min()))]
        if rng.uniform() < 0.5:
            z = float(rng.uniform(0.2, 0.8) * ext[2])
            q = np.array([[0.1 * ext[0], 0.1 * ext[1], z], [0.9 * ext[0], 0.1 * ext[1], z], [0.9 * ext[0], 0.9 * ext[1], z], [0.1 * ext[0], 0.9 * ext[1], z]])
            meshes.append((q, np.array([[0, 1, 2], [0, 2, 3]])))
    ao = int((iso is not None or meshes is not None) and rng.uniform() < 0.4)
    ao_length = float(rng.choice([1e20, 0.3])) * (1.0 if xfm is not None else float(ext.max())) if ao else 1e20
    frames = int(rng.choice([1, 1, 2, 3]))
    kw = dict(W=W, H=H, grad=int(rng.integers(0, 2)), iso=iso, xf=xfs, dt=float(rng.choice([0.5, 0.5, 0.25, 1.0, 0.37])),
              opacity_scale=float(rng.choice([1.0, 0.3, 0.05])), space_skipping=int(rng.uniform() < 0.8),
              frameID=0, camera=camera, xfm=xfm, grad_iso=int(rng.integers(0, 2)),
              multi=bool(fields == 1 or rng.uniform() < 0.7), xf_domains=[(0.0, 1.0)] * fields,
              contour=contour, meshes=meshes, ao=ao, ao_length=ao_length, tf_filter=(0 if rng.uniform() < 0.15 else None))
    desc = dict(seed=seed, scene=what, B=B, levels=levels, root=root, fields=fields, xf=xf_kind, frames=frames,
                xfm=xfm is not None, contour=len(contour or []), meshes=len(meshes or []),
                **{k: v for k, v in kw.items() if k not in ("xf", "camera", "xf_domains", "xfm", "contour", "meshes")})
    return Case(scene, **kw), desc, frames


def random_deep_case(seed):
    """the seventh family: scenes of the C++ generator with 1e5..2e6 cells and thousands of regions — kd trees deep enough
    for the 4-entry short stack to overflow and restart, leaf references packed into the march tree, long rays through
    faint TFs, cameras inside the refined zone."""
    rng = np.random.default_rng(0xDEE9000 + seed)
    kind = str(rng.choice(["lanl", "gear", "exajet"]))
    levels = int(rng.integers(3, 5))
    root = (int(rng.integers(2, 6)), int(rng.integers(2, 4)), int(rng.integers(2, 4)))
    fields = int(rng.choice([1, 1, 2]))
    scene = scenes.generated(kind=kind, seed=int(rng.integers(1, 1 << 20)), root=root, B=8, levels=levels, fields=fields,
                             band=float(rng.choice([1.0, 2.5])))
    ext = np.array(root, dtype=np.float64) * 8 * (1 << (levels - 1))
    centre = 0.5 * ext
    W, H = int(rng.integers(48, 161)), int(rng.integers(32, 121))
    fovy = float(rng.uniform(30.0, 90.0))
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    if rng.uniform() < 0.4:
        camera = (list(rng.uniform(0.2, 0.8, 3) * ext), list(rng.uniform(0.0, 1.0, 3) * ext), [0, 1, 0], fovy)
    else:
        camera = (list(centre + d * float(rng.uniform(0.7, 1.8)) * ext.max()), list(centre + rng.uniform(-0.2, 0.2, 3) * ext), [0, 1, 0], fovy)
    xf_kind = str(rng.choice(["ramp", "band", "table", "steps", "faint", "faint"]))
    xfs = [_random_xf(rng, xf_kind) for _ in range(fields)]
    iso = [(float(rng.uniform(0.2, 0.8)), int(rng.integers(0, fields)))] if rng.uniform() < 0.3 else None
    kw = dict(W=W, H=H, grad=int(rng.integers(0, 2)), iso=iso, xf=xfs, dt=float(rng.choice([0.5, 0.5, 1.0, 0.37])),
              opacity_scale=float(rng.choice([1.0, 0.3, 0.05])), space_skipping=int(rng.uniform() < 0.85), camera=camera,
              grad_iso=int(rng.integers(0, 2)), xf_domains=[(0.0, 1.0)] * fields)
    desc = dict(seed=seed, kind=kind, levels=levels, root=root, fields=fields, xf=xf_kind, cells=int(scene.num_cells),
                **{k: v for k, v in kw.items() if k not in ("xf", "camera", "xf_domains")})
    return Case(scene, **kw), desc
