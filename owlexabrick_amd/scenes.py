"""Scene inputs: the reference's tools/artificial generator restated, and the
ExaBricks in-memory layout (`.bricks` record order, builder/builder.cpp:895-902).

A scene is (bricks7[int32 n,7] = size.xyz, lower.xyz, level; cellIDs[int32];
fields[list of float32 arrays indexed by cellID]).
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np


@dataclass
class Scene:
    bricks7: np.ndarray
    cellIDs: np.ndarray
    fields: List[np.ndarray]
    name: str = "scene"
    value_range: tuple = None  # ScalarField::valueRange of field 0 (includes the loader's 0.0 quirk)
    meta: dict = field(default_factory=dict)

    @property
    def num_cells(self):
        return int(self.cellIDs.size)

    def bounds(self):
        """ExaBricks::getBounds (exa/ExaBricks.cpp:57-63), voxel space."""
        b = self.bricks7.astype(np.int64)
        lo = b[:, 3:6].min(axis=0)
        hi = (b[:, 3:6] + b[:, 0:3] * (1 << b[:, 6:7])).max(axis=0)
        return lo.astype(np.float32), hi.astype(np.float32)


def with_empty_cells(scene, fraction=0.1, seed=0, whole_rows=True):
    """the same scene with some cells missing: their ids become -1, what the reference admits when it is built with
    -DALLOW_EMPTY_CELLS=1 (exa/ExaBricks.cpp:46-49; exaBuilder writes -1 for the holes of a partially filled brick,
    builder/builder.cpp:473-495).  Deterministic in (scene, fraction, seed); whole_rows also knocks out a few complete
    x-rows, so that samples with four and more empty corners occur."""
    rng = np.random.default_rng(0xE3B7 + seed)
    ids = np.array(scene.cellIDs, dtype=np.int32, copy=True)
    ids[rng.uniform(size=ids.size) < fraction] = -1
    if whole_rows:
        begin = 0
        for r in np.asarray(scene.bricks7).reshape(-1, 7):
            sx, sy, sz = int(r[0]), int(r[1]), int(r[2])
            if rng.uniform() < 0.3 and sy > 1:
                y, z = int(rng.integers(sy)), int(rng.integers(sz))
                ids[begin + (z * sy + y) * sx: begin + (z * sy + y) * sx + sx] = -1
            begin += sx * sy * sz
    return Scene(scene.bricks7, ids, scene.fields, name=scene.name + "_holes", value_range=scene.value_range, meta=dict(scene.meta))


def parse_grids(text):
    """tools/artificial/artificial.cpp:140-171: 15-value or 8-value lines; others ignored."""
    grids = []
    for line in text.splitlines():
        tok = line.split()
        vals = None
        try:
            if len(tok) >= 15:
                vals = [int(t, 0) for t in tok[:7]] + [float(t) for t in tok[7:15]]
            elif len(tok) >= 8:
                vals = [int(t, 0) for t in tok[:7]] + [float(tok[7])] * 8
        except ValueError:
            vals = None
        if vals is not None:
            grids.append(vals)
    return grids


def _lerp(v0, v1, x):
    one = np.float32(1.0)
    return (one - x) * v0 + x * v1


def artificial(grids, name="artificial"):
    """exaArtificial restated (tools/artificial/artificial.cpp:78-125): every grid
    line becomes one dense single-level box of cells; cells are emitted x-fastest,
    scalars = trilerp of the 8 corner values at (c-min)/(max-min+1).  Each grid is
    taken as one brick (a valid ExaBricks input: dense, single level), cellIDs =
    running cell index."""
    f32 = np.float32
    bricks, cells, scal = [], [], []
    for g in grids:
        mx, my, mz, nx, ny, nz, lvl = g[:7]
        v = [f32(x) for x in g[7:15]]
        cs = 1 << lvl
        maxc = [mx + (nx - 1) * cs, my + (ny - 1) * cs, mz + (nz - 1) * cs]
        cx = np.arange(mx, maxc[0] + 1, cs)
        cy = np.arange(my, maxc[1] + 1, cs)
        cz = np.arange(mz, maxc[2] + 1, cs)
        # float x = (cx-minCorner[0])/((float)maxCorner[0]-minCorner[0]+1);
        fx = (cx - mx).astype(f32) / f32(f32(maxc[0]) - f32(mx) + f32(1))
        fy = (cy - my).astype(f32) / f32(f32(maxc[1]) - f32(my) + f32(1))
        fz = (cz - mz).astype(f32) / f32(f32(maxc[2]) - f32(mz) + f32(1))
        Z, Y, X = np.meshgrid(fz, fy, fx, indexing="ij")
        s = _lerp(_lerp(_lerp(v[0], v[1], X), _lerp(v[2], v[3], X), Y),
                  _lerp(_lerp(v[4], v[5], X), _lerp(v[6], v[7], X), Y), Z).astype(f32)
        CZ, CY, CX = np.meshgrid(cz, cy, cx, indexing="ij")
        cells.append(np.stack([CX.ravel(), CY.ravel(), CZ.ravel(), np.full(CX.size, lvl)], axis=1).astype(np.int32))
        scal.append(s.ravel())
        bricks.append([nx, ny, nz, mx, my, mz, lvl])
    cells = np.concatenate(cells)
    scal = np.concatenate(scal).astype(f32)
    sc = Scene(np.array(bricks, dtype=np.int32), np.arange(cells.shape[0], dtype=np.int32), [scal], name=name)
    sc.meta["cells"] = cells
    # ScalarField::load sizes the vector by BYTES (exa/ScalarField.cpp:27-34): the
    # zero-filled tail always folds 0.0 into valueRange.
    sc.value_range = (float(min(scal.min(), 0.0)), float(max(scal.max(), 0.0)))
    return sc


EX_GRIDS = {
    # the reference's tools/artificial/ex0..ex4.grids, as data
    "ex0": "0 0 0 1 1 1 0  1.0",
    "ex1": "0 0 0 2 2 2 0  0.0 0.0 0.0 1.0 0.0 0.0 0.0 0.0",
    "ex2": "0 0 0 8 8 8 0  0.0 0.0 0.0 1.0 0.0 0.0 0.0 0.0",
    "ex3": "0 0 0 4 4 4 0  1.0\n4 0 0 2 2 2 1  0.0 0.0 0.0 1.0 0.0 0.0 0.0 0.0\n0 4 0 4 4 4 0  0.75\n4 4 0 4 4 4 0  0.5",
    "ex4": ("0 0 0 4 4 4 0  0.0 0.0 0.0 0.0 1.0 1.0 1.0 1.0\n0 4 0 4 4 4 0  0.8 0.8 0.8 0.8 0.1 0.1 0.1 0.1\n"
            "4 0 0 2 2 2 1  0.0 0.0 0.0 1.0 0.0 0.0 0.0 0.0\n4 4 0 4 4 4 0  0.5"),
    # BASELINE.json configs[0]: 64^3 single-level brick with the ex2 corner pattern
    "c1_64": "0 0 0 64 64 64 0  0.0 0.0 0.0 1.0 0.0 0.0 0.0 0.0",
}


def example(name):
    return artificial(parse_grids(EX_GRIDS[name]), name=name)


def with_extra_field(scene, fn):
    """append a second scalar field computed from cell centres (multi-channel tests)."""
    cells = scene.meta["cells"].astype(np.float32)
    cw = (1 << scene.meta["cells"][:, 3]).astype(np.float32)
    ctr = cells[:, :3] + 0.5 * cw[:, None]
    scene.fields.append(fn(ctr).astype(np.float32))
    return scene


# ---------------------------------------------------------------------------
# procedural block-structured AMR scenes (seeded; stand-ins for the real data
# sets, none of which ship with the reference)
# ---------------------------------------------------------------------------
def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    return z ^ (z >> np.uint64(31))


def _hash01(ix, iy, iz, seed):
    with np.errstate(over="ignore"):
        k = (ix.astype(np.uint64) * np.uint64(0x9E3779B1) ^ iy.astype(np.uint64) * np.uint64(0x85EBCA77)
             ^ iz.astype(np.uint64) * np.uint64(0xC2B2AE3D) ^ np.uint64(seed))
        return (_splitmix64(k) >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def _value_noise(p, cell, seed):
    """trilinear value noise with lattice spacing `cell`."""
    q = p / cell
    i = np.floor(q).astype(np.int64)
    f = q - i
    f = f * f * (3 - 2 * f)
    out = 0.0
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                w = (f[:, 0] if dx else 1 - f[:, 0]) * (f[:, 1] if dy else 1 - f[:, 1]) * (f[:, 2] if dz else 1 - f[:, 2])
                out = out + w * _hash01(i[:, 0] + dx, i[:, 1] + dy, i[:, 2] + dz, seed)
    return out


def amr(seed=1, root=(3, 3, 2), B=4, levels=3, feature="shell", fields=1, name=None):
    """Octree of B^3-cell blocks: root blocks at level levels-1 are refined toward a
    feature surface down to level 0; every leaf block is one brick (dense, single
    level, disjoint).  Voxel space = finest-level cell units."""
    top = levels - 1
    ext = np.array(root, dtype=np.float64) * B * (1 << top)
    centre = ext * np.array([0.45, 0.55, 0.5])
    radius = 0.3 * float(ext.min())

    def sdf(p):
        if feature == "shell":
            return np.abs(np.linalg.norm(p - centre, axis=-1) - radius)
        if feature == "plume":
            d1 = np.abs(np.linalg.norm(p - centre, axis=-1) - radius)
            q = p - centre
            d2 = np.sqrt(q[..., 0] ** 2 + q[..., 2] ** 2) - 0.08 * float(ext.min())
            return np.minimum(d1, np.abs(d2))
        raise ValueError(feature)

    bricks = []
    stack = [(x * B << top, y * B << top, z * B << top, top)
             for z in range(root[2]) for y in range(root[1]) for x in range(root[0])][::-1]
    rng_seed = int(seed)
    while stack:
        x, y, z, l = stack.pop()
        e = B << l
        c = np.array([x + e / 2, y + e / 2, z + e / 2], dtype=np.float64)
        jitter = float(_hash01(np.array([x]), np.array([y]), np.array([z + 7919 * l]), rng_seed)[0])
        if l > 0 and sdf(c) < (0.6 + 0.5 * jitter) * e:
            h = e // 2
            kids = [(x + dx * h, y + dy * h, z + dz * h, l - 1) for dz in (0, 1) for dy in (0, 1) for dx in (0, 1)]
            stack.extend(kids[::-1])
        else:
            bricks.append((B, B, B, x, y, z, l))
    bricks7 = np.array(bricks, dtype=np.int32)
    n = bricks7.shape[0]
    # cell centres in brick order (x fastest)
    k = np.arange(B)
    KZ, KY, KX = np.meshgrid(k, k, k, indexing="ij")
    off = np.stack([KX.ravel(), KY.ravel(), KZ.ravel()], axis=1).astype(np.float64)  # [B^3,3]
    cw = (1 << bricks7[:, 6]).astype(np.float64)
    ctr = (bricks7[:, None, 3:6].astype(np.float64) + (off[None] + 0.5) * cw[:, None, None]).reshape(-1, 3)
    out_fields = []
    for f in range(fields):
        d = np.linalg.norm(ctr - centre, axis=1) / (0.5 * float(np.linalg.norm(ext)))
        base = np.clip(1.0 - d, 0, 1) if f == 0 else 0.5 + 0.5 * np.sin(ctr[:, f % 3] * (6.0 / ext[f % 3]))
        noise = _value_noise(ctr, 0.31 * float(ext.min()), seed + 101 * f) * 0.5 \
            + _value_noise(ctr, 0.13 * float(ext.min()), seed + 101 * f + 17) * 0.25
        out_fields.append((0.55 * base + 0.6 * noise).astype(np.float32))
    # scramble cell IDs so the gather by cellID is exercised (fields are indexed by cellID)
    total = n * B ** 3
    perm = np.argsort(_hash01(np.arange(total), np.zeros(total, dtype=np.int64), np.zeros(total, dtype=np.int64), seed + 5))
    cellIDs = perm.astype(np.int32)
    fields_by_id = []
    for fld in out_fields:
        g = np.empty_like(fld)
        g[cellIDs] = fld
        fields_by_id.append(g)
    sc = Scene(bricks7, cellIDs, fields_by_id, name=name or f"amr_{feature}_{seed}")
    f0 = fields_by_id[0]
    sc.value_range = (float(min(f0.min(), 0.0)), float(max(f0.max(), 0.0)))
    sc.meta.update(dict(B=B, levels=levels, extent=ext.tolist()))
    return sc


# ---------------------------------------------------------------------------
# large seeded scenes from tools/libexa_scenegen.so (C++, multi-threaded)
# ---------------------------------------------------------------------------
KINDS = {"lanl": 0, "gear": 1, "exajet": 2}

# BASELINE.json configs 2..4 as procedural stand-ins; `band` tunes the
# refinement band so that the cell counts land near the targets
CONFIGS = {
    "c2_lanl":   dict(kind="lanl",   seed=0xE7A0001, root=(18, 18, 9), B=8, levels=4, band=1.0, fields=1),
    "c3_gear":   dict(kind="gear",   seed=0xE7A0002, root=(32, 48, 32), B=8, levels=4, band=1.0, fields=2),
    "c4_exajet": dict(kind="exajet", seed=0xE7A0003, root=(64, 32, 32), B=8, levels=4, band=2.5, fields=1),
}


def _scenegen_lib():
    import ctypes as C
    import os
    import subprocess
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    path = os.path.join(here, "libexa_scenegen.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", here, "-s"])
    L = C.CDLL(path)
    L.exa_scenegen_create.argtypes = [C.c_uint64, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                      C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.exa_scenegen_destroy.argtypes = [C.c_void_p]
    for n in ("exa_scenegen_num_bricks", "exa_scenegen_num_cells"):
        getattr(L, n).restype = C.c_uint64
        getattr(L, n).argtypes = [C.c_void_p]
    for n in ("exa_scenegen_bricks7", "exa_scenegen_cell_ids"):
        getattr(L, n).restype = C.c_void_p
        getattr(L, n).argtypes = [C.c_void_p]
    L.exa_scenegen_field.restype = C.c_void_p
    L.exa_scenegen_field.argtypes = [C.c_void_p, C.c_int32]
    L.exa_scenegen_level_histogram.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    return L


class _GenHandle:
    def __init__(self, L, h):
        self.L, self.h = L, h

    def __del__(self):
        if self.h:
            self.L.exa_scenegen_destroy(self.h)
            self.h = None


def generated(kind="exajet", seed=1, root=(8, 4, 4), B=8, levels=4, band=1.0, fields=1, threads=0,
              fill=True, name=None, bricks7=None):
    """seeded AMR scene from the C++ generator; arrays are views on the generator's memory.
    bricks7 ([n,7] int32 size.xyz, lower.xyz, level): the same scene — domain, feature, field functions — on a brick list
    given from outside instead of the generator's own B^3 blocks (e.g. the bricks exaBuilder made of the scene's cells)."""
    import ctypes as C
    L = _scenegen_lib()
    h = C.c_void_p()
    rootN = (C.c_int32 * 3)(*root)
    if bricks7 is not None:
        b7 = np.ascontiguousarray(bricks7, dtype=np.int32).reshape(-1, 7)
        L.exa_scenegen_create_from_bricks.argtypes = [C.c_uint64, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                                      C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
        rc = L.exa_scenegen_create_from_bricks(seed, rootN, B, levels, KINDS[kind], band, fields, threads, b7.ctypes.data,
                                               b7.shape[0], C.byref(h))
    else:
        rc = L.exa_scenegen_create(seed, rootN, B, levels, KINDS[kind], band, fields, threads, int(fill), C.byref(h))
    if rc:
        raise RuntimeError(f"scene generator failed (rc={rc}; more than 2^31 cells?)")
    keep = _GenHandle(L, h)
    nb, nc = L.exa_scenegen_num_bricks(h), L.exa_scenegen_num_cells(h)
    hist = (C.c_uint64 * 8)()
    L.exa_scenegen_level_histogram(h, hist)

    def view(ptr, count, dtype):
        buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype)
    bricks7 = view(L.exa_scenegen_bricks7(h), nb * 7, np.int32).reshape(-1, 7)
    if not fill:
        sc = Scene(bricks7.copy(), np.zeros(0, np.int32), [], name=name or kind)
        sc.meta.update(num_cells=int(nc), levels_hist=list(hist)[:levels])
        return sc
    cellIDs = view(L.exa_scenegen_cell_ids(h), nc, np.int32)
    flds = [view(L.exa_scenegen_field(h, f), nc, np.float32) for f in range(fields)]
    sc = Scene(bricks7, cellIDs, flds, name=name or kind)
    sc.value_range = (0.0, 1.0)
    sc.meta.update(keep=keep, num_cells=int(nc), levels_hist=list(hist)[:levels], kind=kind, seed=seed,
                   root=tuple(root), B=B, levels=levels, band=band)
    return sc


def config(name, scale=1.0, threads=0, fill=True, fields=None, bricks7=None):
    """one of CONFIGS; scale < 1 shrinks the root grid (tests), keeping the feature; `fields` overrides the number
    of scalar fields (C4 with 3 channels as a second data point); `bricks7`: the configuration's scene on an
    external brick list (see generated)."""
    c = dict(CONFIGS[name])
    if fields is not None:
        c["fields"] = int(fields)
    root = tuple(max(1, int(round(r * scale))) for r in c.pop("root"))
    kind = c.pop("kind")
    return generated(kind=kind, root=root, threads=threads, fill=fill, name=name, bricks7=bricks7, **c)


def write_exa(scene, directory, name="scene", remap=None, meshes=None):
    """write the scene in the reference's on-disk formats: `.bricks` (per brick int32
    size[3], lower[3], level, cellIDs[]; builder/builder.cpp:895-902), one raw-float32
    `.scalars` per field in cell-id order (exa/ScalarField.cpp:22-34) and the `.exa`
    config (exa/Config.cpp:88-173).  Returns the config path."""
    import os
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, name + ".bricks"), "wb") as f:
        at = 0
        for rec in np.asarray(scene.bricks7, dtype=np.int32):
            n = int(rec[0]) * int(rec[1]) * int(rec[2])
            f.write(rec.tobytes())
            f.write(np.asarray(scene.cellIDs[at:at + n], dtype=np.int32).tobytes())
            at += n
    lines = ["# written by owlexabrick_amd.scenes.write_exa", f"bricks {name}.bricks"]
    for i, fld in enumerate(scene.fields):
        np.asarray(fld, dtype=np.float32).tofile(os.path.join(directory, f"{name}_{i}.scalars"))
        lines.append(f"scalar field{i} {name}_{i}.scalars")
    if meshes:
        # triangle file: repeated int32 nVerts, vec3f[nVerts], int32 nTris, vec3i[nTris] (exa/TriangleMesh.cpp:21-53)
        with open(os.path.join(directory, name + ".tris"), "wb") as f:
            for verts, tris in meshes:
                v = np.asarray(verts, dtype=np.float32).reshape(-1, 3)
                t = np.asarray(tris, dtype=np.int32).reshape(-1, 3)
                f.write(np.int32(len(v)).tobytes() + v.tobytes() + np.int32(len(t)).tobytes() + t.tobytes())
        lines.append(f"triangles {name}.tris")
    if remap is not None:
        lines.append("remap_from " + " ".join(str(float(v)) for v in remap[0]))
        lines.append("remap_to " + " ".join(str(float(v)) for v in remap[1]))
    path = os.path.join(directory, name + ".exa")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return path
