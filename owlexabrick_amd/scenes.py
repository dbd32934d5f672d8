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
