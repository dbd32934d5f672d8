"""CPU ORACLE (test infrastructure only) for the offline brick builder: a small-case Python
restatement of the reference's builder/builder.cpp, structured like it (id lists, per-slice
statistics, explicit unions per candidate plane).  Only tests import this."""
import numpy as np

SPATIAL_MEDIAN, SAH_ALIKE, SMALL_BRICK_COUNT = 0, 1, 2
IMAX, IMIN = 2 ** 31 - 1, -2 ** 31


def _wrap(v):
    return ((v + 2 ** 31) % 2 ** 32) - 2 ** 31        # two's-complement int32, as the C++ wraps


class Box4:
    def __init__(self):
        self.lo, self.hi = [IMAX] * 4, [IMIN] * 4     # owl box4i(): empty

    def extend(self, o):
        self.lo = [min(a, b) for a, b in zip(self.lo, o.lo)]
        self.hi = [max(a, b) for a, b in zip(self.hi, o.hi)]

    def size(self, k):
        return _wrap(self.hi[k] - self.lo[k])


def cell_bounds(c):                                   # SingleCell::getBounds, builder.cpp:120-130
    b = Box4()
    w = 1 << int(c[3])
    b.lo = [int(c[0]), int(c[1]), int(c[2]), int(c[3])]
    b.hi = [int(c[0]) + w, int(c[1]) + w, int(c[2]) + w, int(c[3]) + 1]
    return b


def unit_cell_volume(b):                              # :163-167
    return b.size(0) * b.size(1) * b.size(2)


def area(b):                                          # :169-176
    return b.size(0) * b.size(1) + b.size(1) * b.size(2) + b.size(2) * b.size(0)


def div_down(a, b):                                   # :44-50 (C division truncates toward zero)
    return a // b if a >= 0 else int((a - (b - 1)) / b)


def div_up(a, b):                                     # :52-58
    return (a + b - 1) // b if a >= 0 else int(a / b)


def all_ids_without_duplicate_cells(cells):           # :301-350
    def key(i):
        c = cells[i]
        u = lambda v: int(v) & 0xFFFFFFFF
        return (u(c[0]) | (u(c[1]) << 32), u(c[2]) | (u(c[3]) << 32), i)
    order = sorted(range(len(cells)), key=key)
    pv = [[tuple(int(x) for x in cells[i]), i] for i in order]
    for i in range(1, len(pv)):
        j = i - 1
        while j >= 0 and pv[j][0][:3] == pv[i][0][:3]:
            if pv[j][0][3] > pv[i][0][3]:
                pv[j] = list(pv[i])
            j -= 1
    out = [pv[0][1]]
    for i in range(1, len(pv)):
        if pv[i][0] != pv[i - 1][0]:
            out.append(pv[i][1])
    return out


def build_bricks(cells, builder_type=SAH_ALIKE, max_leaf_width=127, allow_empty_cells=False):
    """returns [(size3, lower3, level, cellIDs[flat])] in the reference's output order"""
    cells = np.asarray(cells, dtype=np.int64).reshape(-1, 4)
    bricks = []

    def rec(ids):
        # computeCoarsestLevelBounds :185-215
        b = Box4()
        for i in ids:
            b.extend(cell_bounds(cells[i]))
        tight = Box4(); tight.lo, tight.hi = list(b.lo), list(b.hi)        # ALLOW_EMPTY_CELLS: leaf bounds from its cells (:483-495)
        cw = 1 << (b.hi[3] - 1)
        for d in range(3):
            b.lo[d] = cw * div_down(b.lo[d], cw)
            b.hi[d] = cw * div_up(b.hi[d], cw)
        # tryMakeLeaf :447-530
        if (b.size(3) <= 1 and all(b.size(d) // cw <= max_leaf_width for d in range(3))
                and (allow_empty_cells          # ALLOW_EMPTY_CELLS: no volume test (:473-481), holes keep -1
                     or b.size(0) * b.size(1) * b.size(2) * b.size(3) == len(ids) * cw ** 3)):
            if allow_empty_cells:
                b = tight
            sz = [b.size(d) // cw for d in range(3)]
            arr = [-1] * (sz[0] * sz[1] * sz[2])
            for i in ids:
                c = cells[i]
                idx = [(int(c[d]) - b.lo[d]) // cw for d in range(3)]
                arr[idx[0] + sz[0] * (idx[1] + sz[1] * idx[2])] = i
            assert allow_empty_cells or -1 not in arr
            bricks.append((sz, b.lo[:3], b.lo[3], arr))
            return
        dims = [b.size(d) // cw for d in range(3)]
        if dims == [1, 1, 1]:
            raise RuntimeError("coarse size 1 that's not a leaf!?")
        vol = [[0] * dims[d] for d in range(3)]
        sb = [[Box4() for _ in range(dims[d])] for d in range(3)]
        lv = [[[] for _ in range(dims[d])] for d in range(3)]
        for i in ids:                                  # :576-591
            cb = cell_bounds(cells[i])
            for d in range(3):
                s = (cb.lo[d] - b.lo[d]) // cw
                vol[d][s] += unit_cell_volume(cb)
                sb[d][s].extend(cb)
                if int(cells[i][3]) not in lv[d][s]:
                    lv[d][s].append(int(cells[i][3]))
        best_dim, best_pos, best_cost = -1, -1, float("inf")
        if builder_type != SPATIAL_MEDIAN:             # :607-735
            for d in range(3):
                if dims[d] == 0:
                    continue
                expected = unit_cell_volume(b) // dims[d]
                for plane in range(1, dims[d]):
                    L, R = sb[d][plane - 1], sb[d][plane]
                    is_boundary = not (L.lo[3] == R.lo[3] and L.size(3) == R.size(3)
                                       and vol[d][plane - 1] == expected and vol[d][plane] == expected)
                    if not is_boundary:
                        continue
                    lb, rb, ll, rl = Box4(), Box4(), set(), set()
                    for s in range(plane):
                        lb.extend(sb[d][s]); ll |= set(lv[d][s])
                    for s in range(plane, dims[d]):
                        rb.extend(sb[d][s]); rl |= set(lv[d][s])
                    if builder_type == SAH_ALIKE:
                        cost = area(lb) * float(unit_cell_volume(lb)) * lb.size(3) + area(rb) * float(unit_cell_volume(rb)) * rb.size(3)
                    else:
                        cost = float(len(ll)) + float(len(rl))
                    pos = b.lo[d] + plane * cw
                    if cost < best_cost:
                        best_cost, best_dim, best_pos = cost, d, pos
                    elif builder_type == SMALL_BRICK_COUNT and cost == best_cost:
                        middle = dims[best_dim] // 2
                        if abs(pos - middle) < abs(best_pos - middle):
                            best_cost, best_dim, best_pos = cost, d, pos
        if best_dim == -1:                             # :737-744
            best_dim = 0
            for d in (1, 2):
                if abs(dims[d]) > abs(dims[best_dim]):
                    best_dim = d
            best_pos = b.lo[best_dim] + (dims[best_dim] // 2) * cw
        l, r = [], []                                  # :761-778
        for i in ids:
            cb = cell_bounds(cells[i])
            if cb.lo[best_dim] >= best_pos:
                r.append(i)
            elif cb.hi[best_dim] <= best_pos:
                l.append(i)
            else:
                raise RuntimeError("cell straddles split plane!?")
        if not l or not r:
            raise RuntimeError("invalid split...")
        rec(l)                                         # serial_for(2): side 0 = left first (:804-808)
        rec(r)

    rec(all_ids_without_duplicate_cells(cells))
    return bricks


def to_bricks_file_bytes(bricks):
    out = bytearray()
    for sz, lo, lvl, arr in bricks:
        out += np.array(list(sz) + list(lo) + [lvl], dtype=np.int32).tobytes()
        out += np.array(arr, dtype=np.int32).tobytes()
    return bytes(out)
