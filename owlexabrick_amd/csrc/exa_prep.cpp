// exa_prep.cpp — host-side data preparation behind the exa_prep_* C ABI.
//
// What the reference does serially inside OptixRenderer's constructor
// (exa/OptixRenderer.cpp:71-141: brick flattening, index-vector concat, scalar
// gather) and in ExaBrickRegions::buildFrom (exa/Regions.cpp:242-320: overlap
// region partition, finest level, value ranges) is done here with a task-parallel
// partition whose output order is identical to the reference's serial recursion
// (right subtree before left, exa/Regions.cpp:173-178), so region ids, leaf-list
// offsets and every float are the same as the reference would produce.
#include "../../include/exa_hip.h"
#include "exa_ropes.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_prepError;

struct BuildPrim {           // std::pair<box3f,int> of exa/Regions.cpp:75
  float lo[3], hi[3];
  int32_t brickID;
};

// Output of one subtree, in the reference's emission order.
struct RegionChunk {
  std::vector<ExaBrickRegion> regions;   // leafListBegin relative to this chunk
  std::vector<int32_t>        leafList;
  std::vector<ExaKdNode>      nodes;     // the partition's own kd-tree, refs relative to this chunk
  // appends o behind this chunk and returns o's root reference translated into this chunk
  int32_t append(RegionChunk &&o, int32_t oRoot)
  {
    const int32_t base = (int32_t)leafList.size();
    const int32_t regionBase = (int32_t)regions.size(), nodeBase = (int32_t)nodes.size();
    auto fix = [&](int32_t ref) {
      if (ref == EXA_KD_EMPTY) return ref;
      return ref >= 0 ? ref + nodeBase : ~(~ref + regionBase);
    };
    regions.reserve(regions.size() + o.regions.size());
    for (ExaBrickRegion r : o.regions) { r.leafListBegin += base; regions.push_back(r); }
    leafList.insert(leafList.end(), o.leafList.begin(), o.leafList.end());
    nodes.reserve(nodes.size() + o.nodes.size());
    for (ExaKdNode n : o.nodes) { n.left = fix(n.left); n.right = fix(n.right); nodes.push_back(n); }
    return fix(oRoot);
  }
};

struct RegionPartitioner {
  std::atomic<int> workersFree{0};
  static constexpr size_t kSpawnThreshold = 4096;   // prims; below this recurse inline

  // exa/Regions.cpp:32-71 addLeaf
  static int32_t emitLeaf(const std::vector<BuildPrim> &prims, const float lo[3], const float hi[3],
                          RegionChunk &out)
  {
    if (lo[0] >= hi[0] || lo[1] >= hi[1] || lo[2] >= hi[2]) return EXA_KD_EMPTY;
    std::vector<int32_t> ids(prims.size());
    for (size_t i = 0; i < prims.size(); i++) ids[i] = prims[i].brickID;
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());   // std::set<int> order
    if (ids.empty()) return EXA_KD_EMPTY;
    ExaBrickRegion r{};
    for (int k = 0; k < 3; k++) { r.domain_lo[k] = lo[k]; r.domain_hi[k] = hi[k]; }
    r.leafListBegin = (int32_t)out.leafList.size();
    r.leafListSize  = (int32_t)ids.size();
    out.leafList.insert(out.leafList.end(), ids.begin(), ids.end());
    out.regions.push_back(r);
    return ~int32_t(out.regions.size() - 1);
  }

  // exa/Regions.cpp:73-179 buildRec.  Returns the kd reference of this subtree inside `out`:
  // >= 0 node index, < 0 leaf (~region index), EXA_KD_EMPTY nothing.
  int32_t partition(std::vector<BuildPrim> &prims, const float dlo[3], const float dhi[3], RegionChunk &out)
  {
    if (prims.empty()) return EXA_KD_EMPTY;
    for (int i = 0; i < 3; i++) if (dhi[i] == dlo[i]) return EXA_KD_EMPTY;

    // candidate plane per axis: the brick-domain face strictly inside the box that is
    // closest to the box centre; first found wins ties (:84-107)
    float centre[3], bestPos[3], bestDist[3], span[3];
    for (int i = 0; i < 3; i++) {
      span[i] = dhi[i] - dlo[i];
      centre[i] = 0.5f * (dlo[i] + dhi[i]);
      bestPos[i] = dlo[i];
      bestDist[i] = span[i];
    }
    for (const BuildPrim &bp : prims)
      for (int dim = 0; dim < 3; dim++) {
        const float face[2] = { bp.hi[dim], bp.lo[dim] };        // side 0 = upper, side 1 = lower
        for (int side = 0; side < 2; side++) {
          const float pos = face[side];
          if (pos <= dlo[dim] || pos >= dhi[dim]) continue;
          const float dist = std::fabs(centre[dim] - pos);
          if (dist >= bestDist[dim]) continue;
          bestPos[dim] = pos;
          bestDist[dim] = dist;
        }
      }
    int widest = 0;                                                // arg_max(span) (:112)
    for (int i = 1; i < 3; i++) if (std::fabs(span[i]) > std::fabs(span[widest])) widest = i;
    int splitDim = -1;
    float splitPos = 0.f;
    for (int i = 0; i < 3; i++) {                                  // (:113-123)
      const int dim = (widest + i) % 3;
      if (bestPos[dim] <= dlo[dim] || bestPos[dim] >= dhi[dim]) continue;
      splitDim = dim;
      splitPos = bestPos[dim];
      break;
    }
    if (splitDim < 0) return emitLeaf(prims, dlo, dhi, out);        // (:131-134)

    float llo[3], lhi[3], rlo[3], rhi[3];
    for (int k = 0; k < 3; k++) { llo[k] = rlo[k] = dlo[k]; lhi[k] = rhi[k] = dhi[k]; }
    lhi[splitDim] = splitPos;
    rlo[splitDim] = splitPos;
    std::vector<BuildPrim> left, right;
    left.reserve(prims.size() / 2 + 8);
    right.reserve(prims.size() / 2 + 8);
    for (const BuildPrim &bp : prims) {                            // clip to both halves (:142-169)
      BuildPrim c = bp;
      // only splitDim changes: intersection() with a half-box that shares every other face
      c.hi[splitDim] = std::fmin(bp.hi[splitDim], splitPos);
      if (c.lo[0] < c.hi[0] && c.lo[1] < c.hi[1] && c.lo[2] < c.hi[2]) left.push_back(c);
      c.hi[splitDim] = bp.hi[splitDim];
      c.lo[splitDim] = std::fmax(bp.lo[splitDim], splitPos);
      if (c.lo[0] < c.hi[0] && c.lo[1] < c.hi[1] && c.lo[2] < c.hi[2]) right.push_back(c);
    }
    std::vector<BuildPrim>().swap(prims);                          // buildPrims.clear() (:171)

    // right subtree first, then left (:173-178).  Big subtrees run as tasks; their
    // chunks are concatenated in that same order.
    const bool spawn = left.size() >= kSpawnThreshold && right.size() >= kSpawnThreshold
                       && workersFree.fetch_sub(1) > 0;
    const int32_t me = (int32_t)out.nodes.size();
    out.nodes.push_back(ExaKdNode{ splitPos, splitDim, EXA_KD_EMPTY, EXA_KD_EMPTY });
    int32_t leftRef, rightRef;
    if (spawn) {
      RegionChunk rightOut, leftOut;
      int32_t rr = EXA_KD_EMPTY, lr = EXA_KD_EMPTY;
      std::thread t([&] { rr = partition(right, rlo, rhi, rightOut); });
      lr = partition(left, llo, lhi, leftOut);
      t.join();
      workersFree.fetch_add(1);
      rightRef = out.append(std::move(rightOut), rr);
      leftRef = out.append(std::move(leftOut), lr);
    } else {
      if (left.size() >= kSpawnThreshold && right.size() >= kSpawnThreshold) workersFree.fetch_add(1);
      rightRef = partition(right, rlo, rhi, out);
      leftRef = partition(left, llo, lhi, out);
    }
    out.nodes[me].left = leftRef;      // lower side of the split plane
    out.nodes[me].right = rightRef;    // upper side
    return me;
  }
};

template <typename F>
void parallelChunks(size_t n, int nthreads, F &&body)
{
  if (n == 0) return;
  nthreads = std::max(1, std::min<int>(nthreads, (int)std::min<size_t>(n, 1024)));
  if (nthreads == 1) { body(0, n); return; }
  std::atomic<size_t> next{0};
  const size_t grain = std::max<size_t>(1, n / (size_t(nthreads) * 16));
  std::vector<std::thread> pool;
  for (int t = 0; t < nthreads; t++)
    pool.emplace_back([&] {
      for (;;) {
        const size_t b = next.fetch_add(grain);
        if (b >= n) break;
        body(b, std::min(n, b + grain));
      }
    });
  for (auto &t : pool) t.join();
}

} // namespace

struct ExaPrep {
  std::vector<ExaBrick>       bricks;
  std::vector<ExaBrickRegion> regions;
  std::vector<int32_t>        leafList;
  std::vector<ExaKdNode>      kdNodes;
  int32_t                     kdRoot = EXA_KD_EMPTY;
  std::unique_ptr<float[]>    scalars;        // numFields * totalCells, first touched in parallel
  std::vector<uint64_t>       channelOffset;
  uint64_t totalCells = 0;
  int32_t  numFields = 0;
  int32_t  allowEmptyCells = 0;
  float    boundsLo[3], boundsHi[3];
};

namespace {

// exa/Regions.cpp:182-240 computeValueRange.  Cell hat supports are monotone in
// the cell index, so the per-axis "touches the region" flags of the reference form
// one contiguous index range per axis; only that sub-box is scanned.
void valueRangeOf(const ExaPrep &P, ExaBrickRegion &R, int numRegionFields)
{
  float lo = std::numeric_limits<float>::infinity(), hi = -lo;
  for (int f = 0; f < numRegionFields; f++) {
    const float *field = P.scalars.get() + P.channelOffset[f];
    for (int i = 0; i < R.leafListSize; i++) {
      const ExaBrick &b = P.bricks[P.leafList[R.leafListBegin + i]];
      const float cw = float(1 << b.level);
      int r0[3], r1[3];
      bool empty = false;
      for (int k = 0; k < 3; k++) {
        int first = -1, last = -2;
        for (int c = 0; c < b.size[k]; c++) {
          const float pos = b.lower[k] + (c + .5f) * cw;
          const bool valid = (pos - cw <= R.domain_hi[k]) && (pos + cw >= R.domain_lo[k]);
          if (valid) { if (first < 0) first = c; last = c; }
        }
        r0[k] = first; r1[k] = last;
        if (first < 0) empty = true;
      }
      if (empty) continue;
      for (int iz = r0[2]; iz <= r1[2]; iz++)
        for (int iy = r0[1]; iy <= r1[1]; iy++) {
          const float *row = field + b.begin + size_t(b.size[0]) * iy + size_t(b.size[0]) * b.size[1] * iz;
          for (int ix = r0[0]; ix <= r1[0]; ix++) {
            const float s = row[ix];
            if (s < lo) lo = s;
            if (s > hi) hi = s;
          }
        }
    }
  }
  R.valueRange_lo = lo;
  R.valueRange_hi = hi;
}

} // namespace

extern "C" {

const char *exa_prep_last_error(void) { return g_prepError.c_str(); }

int exa_prep_create(const int32_t *bricks7, uint64_t numBricks,
                    const int32_t *cellIDs, uint64_t numCellIDs,
                    const float *const *fields, const uint64_t *fieldLen,
                    int32_t numFields, int32_t numRegionFields, int32_t numThreads,
                    ExaPrep **out)
{ return exa_prep_create_ex(bricks7, numBricks, cellIDs, numCellIDs, fields, fieldLen, numFields, numRegionFields, numThreads, 0, out); }

int exa_prep_create_ex(const int32_t *bricks7, uint64_t numBricks,
                       const int32_t *cellIDs, uint64_t numCellIDs,
                       const float *const *fields, const uint64_t *fieldLen,
                       int32_t numFields, int32_t numRegionFields, int32_t numThreads, int32_t flags,
                       ExaPrep **out)
{
  if (!out) return 1;
  *out = nullptr;
  if (numThreads <= 0) numThreads = (int)std::max(1u, std::thread::hardware_concurrency());
  if (numRegionFields < 0 || numRegionFields > numFields) numRegionFields = numFields;
  ExaPrep *P = new ExaPrep;
  const bool verbose = std::getenv("EXA_PREP_VERBOSE") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tPhase = now();
  auto lap = [&](const char *what) { if (verbose) { const double t = now(); std::fprintf(stderr, "[exa_prep] %-10s %.3f s\n", what, t - tPhase); tPhase = t; } };
  auto fail = [&](const char *msg) { g_prepError = msg; delete P; return 1; };

  // ---- flatten (exa/OptixRenderer.cpp:71-93) ----
  P->bricks.resize(numBricks);
  P->numFields = numFields;
  P->allowEmptyCells = (flags & EXA_PREP_ALLOW_EMPTY_CELLS) ? 1 : 0;
  const bool allowEmpty = P->allowEmptyCells != 0;
  for (int k = 0; k < 3; k++) { P->boundsLo[k] = INFINITY; P->boundsHi[k] = -INFINITY; }
  uint64_t running = 0;
  for (uint64_t i = 0; i < numBricks; i++) {
    const int32_t *r = bricks7 + 7 * i;
    ExaBrick &b = P->bricks[i];
    b.size[0] = r[0]; b.size[1] = r[1]; b.size[2] = r[2];
    b.lower[0] = r[3]; b.lower[1] = r[4]; b.lower[2] = r[5];
    b.level = r[6];
    if (running > 0x7fffffffull) return fail("32-bit offset overflow");
    b.begin = (uint32_t)running;
    running += uint64_t(b.size[0]) * uint64_t(b.size[1]) * uint64_t(b.size[2]);
    if (running > numCellIDs) return fail("failed sanity-check in brick size");
    for (int k = 0; k < 3; k++) {                 // ExaBricks::getBounds (exa/ExaBricks.cpp:57-63)
      P->boundsLo[k] = std::fmin(P->boundsLo[k], float(b.lower[k]));
      P->boundsHi[k] = std::fmax(P->boundsHi[k], float(b.lower[k] + b.size[k] * (1 << b.level)));
    }
  }
  if (running != numCellIDs) return fail("failed sanity-check in brick size");
  P->totalCells = running;

  lap("flatten");
  // ---- gather scalars into brick order (exa/OptixRenderer.cpp:103-132) ----
  P->scalars.reset(new float[size_t(numFields) * P->totalCells]);
  P->channelOffset.resize(numFields);
  std::atomic<int> bad{0};
  for (int f = 0; f < numFields; f++) {
    P->channelOffset[f] = uint64_t(f) * P->totalCells;
    float *dst = P->scalars.get() + P->channelOffset[f];
    const float *src = fields[f];
    const uint64_t len = fieldLen[f];
    parallelChunks(P->totalCells, numThreads, [&](size_t b, size_t e) {
      for (size_t i = b; i < e; i++) {
        const int32_t id = cellIDs[i];
        if (id < 0) {
          // ALLOW_EMPTY_CELLS: the renderer poisons EVERY negative id (exa/OptixRenderer.cpp:116-118).  That only -1 means
          // "no cell" is an assert of the loader (exa/ExaBricks.cpp:46-49) — compiled out of a release build, so a bricks
          // file the reference's release binary renders is rendered here as well.
          if (allowEmpty) dst[i] = EXA_EMPTY_CELL_POISON_VALUE; else bad = 1;
          continue;
        }
        if (uint64_t(id) >= len) { bad = 2; continue; }
        dst[i] = src[id];
      }
    });
  }
  if (bad == 1) return fail("overflow in index vector...");
  if (bad == 2) return fail("invalid cell ID");

  lap("gather");
  // ---- same-bricks regions (exa/Regions.cpp:242-320) ----
  std::vector<BuildPrim> prims(numBricks);
  float blo[3] = { INFINITY, INFINITY, INFINITY }, bhi[3] = { -INFINITY, -INFINITY, -INFINITY };
  for (uint64_t i = 0; i < numBricks; i++) {
    const ExaBrick &b = P->bricks[i];
    const float cw = float(1 << b.level);        // Brick::getDomain (programs/Brick.h:50-55)
    for (int k = 0; k < 3; k++) {
      prims[i].lo[k] = float(b.lower[k]) - 0.5f * cw;
      prims[i].hi[k] = float(b.lower[k]) + (float(b.size[k]) + 0.5f) * cw;
      blo[k] = std::fmin(blo[k], prims[i].lo[k]);
      bhi[k] = std::fmax(bhi[k], prims[i].hi[k]);
    }
    prims[i].brickID = (int32_t)i;
  }
  RegionPartitioner part;
  part.workersFree = numThreads - 1;
  RegionChunk all;
  P->kdRoot = part.partition(prims, blo, bhi, all);
  P->regions = std::move(all.regions);
  P->leafList = std::move(all.leafList);
  P->kdNodes = std::move(all.nodes);

  lap("partition");
  // finest level + value range per region (exa/Regions.cpp:290-306)
  parallelChunks(P->regions.size(), numThreads, [&](size_t b, size_t e) {
    for (size_t r = b; r < e; r++) {
      ExaBrickRegion &R = P->regions[r];
      int finest = 1 << 30;
      for (int i = 0; i < R.leafListSize; i++)
        finest = std::min(finest, P->bricks[P->leafList[R.leafListBegin + i]].level);
      R.finestLevelCellWidth = float(1 << finest);
      valueRangeOf(*P, R, numRegionFields);
    }
  });
  lap("ranges");
  *out = P;
  return 0;
}

void exa_prep_destroy(ExaPrep *P) { delete P; }

// Diagnostic (tests/test_ropes.py): the leaves and links exa_hip builds for its rope walk from this scene's kd-tree, on the
// host.  First call with leaves == NULL to learn the counts, then with arrays of that size:
// leafBoxes numLeaves x 6 floats (lo, hi), leafLinks numLeaves x 6 (-x +x -y +y -z +z: inner node >= 0, leaf ~i, outside =
// EXA_KD_EMPTY + 1), leafRegion numLeaves (region id, -1 for a gap), nodes numNodes (no empty slots).  flags out: bit 0 the
// boxes equal the region domains, bit 1 the planes are in the short division's range.
int exa_prep_ropes(const ExaPrep *P, uint64_t *numLeaves, uint64_t *numNodes, float *leafBoxes, int32_t *leafLinks, int32_t *leafRegion,
                   ExaKdNode *nodes, int32_t *flags)
{
  if (!P || !numLeaves || !numNodes) return 1;
  std::vector<float> dom(6 * P->regions.size());
  float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
  for (size_t r = 0; r < P->regions.size(); r++)
    for (int k = 0; k < 3; k++) {
      dom[6 * r + k] = P->regions[r].domain_lo[k]; dom[6 * r + 3 + k] = P->regions[r].domain_hi[k];
      lo[k] = std::fmin(lo[k], P->regions[r].domain_lo[k]); hi[k] = std::fmax(hi[k], P->regions[r].domain_hi[k]);
    }
  exa::RopeBuild rb;
  exa::buildRopesHost(P->kdNodes.data(), P->kdNodes.size(), P->kdRoot, dom.data(), P->regions.size(), lo, hi,
                      std::min(16u, std::max(1u, std::thread::hardware_concurrency())), rb);
  if (!leafBoxes) { *numLeaves = rb.leaves.size(); *numNodes = rb.nodes.size(); return 0; }
  if (*numLeaves != rb.leaves.size() || *numNodes != rb.nodes.size()) { g_prepError = "exa_prep_ropes: array sizes do not match the counts"; return 1; }
  for (size_t i = 0; i < rb.leaves.size(); i++) {
    for (int k = 0; k < 3; k++) { leafBoxes[6 * i + k] = rb.leaves[i].lo[k]; leafBoxes[6 * i + 3 + k] = rb.leaves[i].hi[k]; }
    for (int f = 0; f < 6; f++) leafLinks[6 * i + f] = rb.leaves[i].rope[f];
    leafRegion[i] = rb.leaves[i].region;
  }
  if (nodes) std::copy(rb.nodes.begin(), rb.nodes.end(), nodes);
  if (flags) *flags = (rb.boxesMatch ? 1 : 0) | (rb.planesOnGrid ? 2 : 0);
  return 0;
}

int exa_prep_scene(const ExaPrep *P, ExaHipScene *out)
{
  if (!P || !out) return 1;
  out->bricks = P->bricks.data();               out->numBricks = P->bricks.size();
  out->regions = P->regions.data();             out->numRegions = P->regions.size();
  out->leafList = P->leafList.data();           out->leafListSize = P->leafList.size();
  out->scalars = P->scalars.get();
  out->channelOffset = P->channelOffset.data();
  out->totalCells = P->totalCells;
  out->numFields = P->numFields;
  out->kdNodes = P->kdNodes.data();
  out->numKdNodes = P->kdNodes.size();
  out->kdRoot = P->kdRoot;
  out->allowEmptyCells = P->allowEmptyCells;
  for (int k = 0; k < 3; k++) { out->voxelBounds_lo[k] = P->boundsLo[k]; out->voxelBounds_hi[k] = P->boundsHi[k]; }
  return 0;
}

} // extern "C"
