// exa_builder.cpp — exaBuilder: cells -> bricks (`.cells` -> `.bricks`), the offline step in front
// of the renderer (builder/builder.cpp of the reference: same command line, same output file).
//
//   exaBuilder in.cells -o out.bricks [--max-leaf-width N] [--spatial-median] [--large-bricks]
//              [--parallel] [-v]
//
// Algorithm as in the reference: drop duplicate cells (finer wins, builder.cpp:301-350), then split
// the cell set recursively on planes of the coarsest-level grid until a node is one dense
// single-level box of at most maxLeafWidth cells per axis (tryMakeLeaf, :447-530).  Split choice
// (:538-735): among planes where the two adjacent slices differ (level range or fullness) take the
// cheapest under the SAH-alike cost (or the fewest-levels cost with --large-bricks), else the
// spatial median of the widest axis.  Here the cell ids are partitioned in place in one index
// array and slice statistics use prefix/suffix unions; bricks come out in the reference's order
// (left subtree first), so the `.bricks` file is byte-identical.
#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

enum BuilderType { SPATIAL_MEDIAN = 0, SAH_ALIKE = 1, SMALL_BRICK_COUNT = 2 };

struct Cell { int32_t x, y, z, level; };

// 4-d integer box (x,y,z,level) with the reference's empty-box convention; differences wrap like
// two's-complement ints (the reference subtracts INT_MAX from INT_MIN for an empty slice)
struct Box4 {
  int32_t lo[4] = { INT_MAX, INT_MAX, INT_MAX, INT_MAX }, hi[4] = { INT_MIN, INT_MIN, INT_MIN, INT_MIN };
  void extend(const Box4 &b) { for (int k = 0; k < 4; k++) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
  int32_t size(int k) const { return (int32_t)((uint32_t)hi[k] - (uint32_t)lo[k]); }
};
inline Box4 boundsOf(const Cell &c)
{
  Box4 b;
  const int w = 1 << c.level;
  b.lo[0] = c.x; b.lo[1] = c.y; b.lo[2] = c.z; b.lo[3] = c.level;
  b.hi[0] = c.x + w; b.hi[1] = c.y + w; b.hi[2] = c.z + w; b.hi[3] = c.level + 1;
  return b;
}
inline uint64_t unitCellVolume(const Box4 &b) { return uint64_t(int64_t(b.size(0))) * uint64_t(int64_t(b.size(1))) * uint64_t(int64_t(b.size(2))); }
inline uint64_t area(const Box4 &b)
{
  const int64_t x = b.size(0), y = b.size(1), z = b.size(2);
  return uint64_t(x) * uint64_t(y) + uint64_t(y) * uint64_t(z) + uint64_t(z) * uint64_t(x);
}
inline int divDown(int a, int b) { return a >= 0 ? a / b : (a - (b - 1)) / b; }
inline int divUp(int a, int b) { return a >= 0 ? (a + b - 1) / b : a / b; }

struct Builder {
  const std::vector<Cell> &cells;
  std::vector<int32_t> ids;
  int type, maxLeafWidth;
  bool allowEmptyCells = false;   // the reference's build option ALLOW_EMPTY_CELLS (builder.cpp:473-495), here --allow-empty-cells
  bool verbose;
  std::vector<int32_t> bricks7, brickCells;       // output, `.bricks` order

  Builder(const std::vector<Cell> &c, int type, int maxLeafWidth, bool verbose)
    : cells(c), type(type), maxLeafWidth(maxLeafWidth), verbose(verbose) {}

  // builder.cpp:301-350: sort by the raw 64-bit words of the cell record, let a finer cell overwrite
  // a coarser one at the same position when they are neighbours in that order, drop repeats
  void initialIds()
  {
    struct Keyed { uint64_t k0, k1; int32_t id; Cell c; };
    std::vector<Keyed> v(cells.size());
    for (size_t i = 0; i < cells.size(); i++) {
      const Cell &c = cells[i];
      v[i] = { uint64_t(uint32_t(c.x)) | uint64_t(uint32_t(c.y)) << 32, uint64_t(uint32_t(c.z)) | uint64_t(uint32_t(c.level)) << 32, int32_t(i), c };
    }
    std::sort(v.begin(), v.end(), [](const Keyed &a, const Keyed &b) {
      if (a.k0 != b.k0) return a.k0 < b.k0;
      if (a.k1 != b.k1) return a.k1 < b.k1;
      return a.id < b.id;
    });
    auto samePos = [](const Keyed &a, const Keyed &b) { return a.c.x == b.c.x && a.c.y == b.c.y && a.c.z == b.c.z; };
    for (size_t i = 1; i < v.size(); i++)
      for (size_t j = i; j-- > 0 && samePos(v[j], v[i]);)
        if (v[j].c.level > v[i].c.level) v[j] = v[i];
    ids.clear();
    if (v.empty()) return;
    ids.push_back(v[0].id);
    for (size_t i = 1; i < v.size(); i++)
      if (v[i].k0 != v[i - 1].k0 || v[i].k1 != v[i - 1].k1) ids.push_back(v[i].id);
  }

  void emitBrick(const Box4 &b, size_t lo, size_t hi)
  {
    const int cw = 1 << b.lo[3];
    const int sx = b.size(0) / cw, sy = b.size(1) / cw, sz = b.size(2) / cw;
    const int32_t rec[7] = { sx, sy, sz, b.lo[0], b.lo[1], b.lo[2], b.lo[3] };
    bricks7.insert(bricks7.end(), rec, rec + 7);
    const size_t at = brickCells.size();
    brickCells.resize(at + size_t(sx) * sy * sz, -1);
    for (size_t i = lo; i < hi; i++) {
      const Cell &c = cells[ids[i]];
      const size_t idx = size_t((c.x - b.lo[0]) / cw) + size_t(sx) * (size_t((c.y - b.lo[1]) / cw) + size_t(sy) * size_t((c.z - b.lo[2]) / cw));
      brickCells[at + idx] = ids[i];
    }
  }

  void build(size_t lo, size_t hi)
  {
    // coarse-aligned bounds of this cell set (computeCoarsestLevelBounds, :185-215)
    Box4 b;
    for (size_t i = lo; i < hi; i++) b.extend(boundsOf(cells[ids[i]]));
    const Box4 tight = b;            // ALLOW_EMPTY_CELLS: a leaf's bounds are rebuilt from its cells (:483-495)
    const int cw = 1 << (b.hi[3] - 1);
    for (int d = 0; d < 3; d++) { b.lo[d] = cw * divDown(b.lo[d], cw); b.hi[d] = cw * divUp(b.hi[d], cw); }
    const size_t n = hi - lo;
    // tryMakeLeaf (:447-530)
    if (b.size(3) <= 1 && b.size(0) / cw <= maxLeafWidth && b.size(1) / cw <= maxLeafWidth && b.size(2) / cw <= maxLeafWidth
        && (allowEmptyCells        // ALLOW_EMPTY_CELLS: partially filled bricks pass, their holes keep the id -1 (:473-495)
            || uint64_t(int64_t(b.size(0))) * uint64_t(int64_t(b.size(1))) * uint64_t(int64_t(b.size(2))) * uint64_t(int64_t(b.size(3)))
                   == uint64_t(n) * uint64_t(cw) * cw * cw)) {
      emitBrick(allowEmptyCells ? tight : b, lo, hi);
      return;
    }
    const int dims[3] = { b.size(0) / cw, b.size(1) / cw, b.size(2) / cw };
    if (dims[0] == 1 && dims[1] == 1 && dims[2] == 1) throw std::runtime_error("coarse size 1 that's not a leaf!?");

    // per-axis slices of the coarse grid: occupied volume, 4-d bounds, set of levels (:560-596)
    std::vector<uint64_t> vol[3];
    std::vector<Box4> sb[3];
    std::vector<uint32_t> lv[3];
    for (int d = 0; d < 3; d++) { vol[d].assign(dims[d], 0); sb[d].assign(dims[d], Box4()); lv[d].assign(dims[d], 0u); }
    for (size_t i = lo; i < hi; i++) {
      const Cell &c = cells[ids[i]];
      const Box4 cb = boundsOf(c);
      const int bin[3] = { (c.x - b.lo[0]) / cw, (c.y - b.lo[1]) / cw, (c.z - b.lo[2]) / cw };
      for (int d = 0; d < 3; d++) { vol[d][bin[d]] += unitCellVolume(cb); sb[d][bin[d]].extend(cb); lv[d][bin[d]] |= 1u << c.level; }
    }

    int bestDim = -1, bestPos = -1;
    double bestCost = std::numeric_limits<float>::infinity();
    if (type != SPATIAL_MEDIAN) {
      for (int d = 0; d < 3; d++) {
        if (dims[d] == 0) continue;
        const uint64_t full = unitCellVolume(b) / uint64_t(dims[d]);
        // unions of everything left of / right of each plane
        std::vector<Box4> pre(dims[d] + 1), suf(dims[d] + 1);
        std::vector<uint32_t> preL(dims[d] + 1, 0u), sufL(dims[d] + 1, 0u);
        for (int s = 0; s < dims[d]; s++) { pre[s + 1] = pre[s]; pre[s + 1].extend(sb[d][s]); preL[s + 1] = preL[s] | lv[d][s]; }
        for (int s = dims[d]; s-- > 0;) { suf[s] = suf[s + 1]; suf[s].extend(sb[d][s]); sufL[s] = sufL[s + 1] | lv[d][s]; }
        for (int plane = 1; plane < dims[d]; plane++) {
          const Box4 &L = sb[d][plane - 1], &R = sb[d][plane];
          const bool boundary = !(L.lo[3] == R.lo[3] && L.size(3) == R.size(3) && vol[d][plane - 1] == full && vol[d][plane] == full);
          if (!boundary) continue;
          const Box4 &lb = pre[plane], &rb = suf[plane];
          double cost;
          if (type == SAH_ALIKE)
            cost = area(lb) * (double)unitCellVolume(lb) * lb.size(3) + area(rb) * (double)unitCellVolume(rb) * rb.size(3);
          else
            cost = (double)__builtin_popcount(preL[plane]) + (double)__builtin_popcount(sufL[plane]);
          const int pos = b.lo[d] + plane * cw;
          if (cost < bestCost) { bestCost = cost; bestDim = d; bestPos = pos; }
          else if (type == SMALL_BRICK_COUNT && cost == bestCost) {
            const int middle = dims[bestDim] / 2;                    // (:722-730, as written there)
            if (std::abs(pos - middle) < std::abs(bestPos - middle)) { bestCost = cost; bestDim = d; bestPos = pos; }
          }
        }
      }
    }
    if (bestDim < 0) {                                             // spatial median of the widest axis (:737-744)
      bestDim = 0;
      for (int d = 1; d < 3; d++) if (std::abs(dims[d]) > std::abs(dims[bestDim])) bestDim = d;
      bestPos = b.lo[bestDim] + (dims[bestDim] / 2) * cw;
    }
    // partition in place: left = cells entirely below the plane
    auto coord = [&](const Cell &c) { return bestDim == 0 ? c.x : (bestDim == 1 ? c.y : c.z); };
    size_t mid = lo;
    for (size_t i = lo; i < hi; i++) {
      const Cell &c = cells[ids[i]];
      const int cl = coord(c), ch = cl + (1 << c.level);
      if (cl >= bestPos) continue;
      if (ch > bestPos) throw std::runtime_error("cell straddles split plane!?");
      std::swap(ids[i], ids[mid++]);
    }
    if (mid == lo || mid == hi) throw std::runtime_error("invalid split...");
    build(lo, mid);
    build(mid, hi);
  }
};

} // namespace

int main(int argc, char **argv)
{
  try {
    bool spatialMedian = false, largeBricks = false, verbose = false, allowEmptyCells = false;
    int maxLeafWidth = 127;
    std::string in, out;
    for (int i = 1; i < argc; i++) {
      const std::string a = argv[i];
      if (a[0] != '-') in = a;
      else if (a == "-o" && i + 1 < argc) out = argv[++i];
      else if (a == "-kd" && i + 1 < argc) ++i;                    // kd dump is never read by the renderer
      else if (a == "--parallel") {}
      else if (a == "--max-leaf-width" && i + 1 < argc) maxLeafWidth = std::stoi(argv[++i]);
      else if (a == "-v") verbose = true;
      else if (a == "--allow-empty-cells") allowEmptyCells = true;
      else if (a == "--no-shift-planes" || a == "--no-planes" || a == "--spatial-median" || a == "--spatial-median-builder") spatialMedian = true;
      else if (a == "--large-bricks") largeBricks = true;
      else throw std::runtime_error("un-recognized cmdline arg '" + a + "'");
    }
    if (in.empty()) throw std::runtime_error("no input file specified...");
    if (out.empty()) throw std::runtime_error("no output file specified...");
    if (largeBricks && spatialMedian) throw std::runtime_error("you gotta decide, either spatial median _or_ large bricks...");
    std::ifstream f(in, std::ios::binary);
    if (!f.good()) throw std::runtime_error("could not open " + in);
    f.seekg(0, f.end);
    const size_t n = size_t(f.tellg()) / sizeof(Cell);              // `.cells`: int32 x,y,z,level per cell (:813-834)
    f.seekg(0, f.beg);
    std::vector<Cell> cells(n);
    f.read(reinterpret_cast<char *>(cells.data()), std::streamsize(n * sizeof(Cell)));
    for (const Cell &c : cells) if (c.level < 0 || c.level > 30) throw std::runtime_error("cell level out of range");
    const int type = (!spatialMedian && !largeBricks) ? SAH_ALIKE : (largeBricks ? SMALL_BRICK_COUNT : SPATIAL_MEDIAN);
    Builder b(cells, type, maxLeafWidth, verbose);
    b.allowEmptyCells = allowEmptyCells;
    b.initialIds();
    if (b.ids.empty()) throw std::runtime_error("no cells");
    b.build(0, b.ids.size());
    std::cout << "Done bricking, created " << b.bricks7.size() / 7 << " bricks" << std::endl;
    std::ofstream o(out, std::ios::binary);
    size_t at = 0;
    for (size_t i = 0; i < b.bricks7.size() / 7; i++) {
      const int32_t *r = &b.bricks7[7 * i];
      const size_t m = size_t(r[0]) * r[1] * r[2];
      o.write(reinterpret_cast<const char *>(r), 7 * sizeof(int32_t));
      o.write(reinterpret_cast<const char *>(b.brickCells.data() + at), std::streamsize(m * sizeof(int32_t)));
      at += m;
    }
    return 0;
  } catch (const std::exception &e) {
    std::cerr << "FATAL Error : " << e.what() << std::endl;
    return 1;
  }
}
