// exa_module.cpp — the exa_hip_* C ABI: scene upload, LBVH topology build,
// dirty-flag activity/refit, frame launch.  Host side of what
// exa/OptixRenderer.cpp does through OWL/OptiX; see include/exa_hip.h for the
// per-entry citations.
#include "exa_device.h"
#include "exa_ropes.h"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>
#include <utility>
#include <vector>

using namespace exa;

// calibration of the wide-march assignment (assignWide), overridable for A/B builds
#ifndef EXA_WIDE_SPEED2
#define EXA_WIDE_SPEED2 1.63
#endif
#ifndef EXA_WIDE_WORK2
#define EXA_WIDE_WORK2 1.49
#endif
#ifndef EXA_WIDE_WORK4
#define EXA_WIDE_WORK4 1.88
#endif

// the launchers of the sampling kernels exist once per association of the basis sums (exa_device.h)
#define EXA_FORM(fn) (emptyCells ? form0e::fn : (basisForm ? form1::fn : form0::fn))

namespace {

thread_local std::string g_createError;

#define HIP_TRY(h, call)                                                                   \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      (h)->fail(std::string(#call) + ": " + hipGetErrorString(e_));                        \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)

// ---- LBVH topology: Morton-sorted regions, split at the highest differing bit;
// the depth is capped at kStackDepth so the per-lane LDS stack cannot overflow
// (median splits once the remaining depth budget is tight). ----
struct LbvhTopology {
  std::vector<int32_t> child0, child1;
  std::vector<int32_t> height;          // per internal node
  std::vector<uint64_t> codes;
  std::vector<uint32_t> order;

  static uint64_t spread21(uint64_t v)
  {
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
  }
  static int ceilLog2(uint64_t n) { int l = 0; while ((1ull << l) < n) l++; return l; }

  int32_t buildRange(size_t lo, size_t hi, int depth, int32_t &outHeight)
  {
    if (hi - lo == 1) { outHeight = 0; return ~int32_t(order[lo]); }
    const int32_t me = (int32_t)child0.size();
    child0.push_back(0); child1.push_back(0); height.push_back(0);
    const size_t n = hi - lo;
    size_t split = lo + (n + 1) / 2;                           // median fallback
    const uint64_t first = codes[lo], last = codes[hi - 1];
    if (first != last) {
      const int prefix = __builtin_clzll(first ^ last);
      // Karras-style search: last index whose code shares more than `prefix` leading
      // bits with `first`; the right child starts right after it
      size_t at = lo, step = hi - 1 - lo;
      do {
        step = (step + 1) >> 1;
        const size_t cand = at + step;
        if (cand < hi - 1) {
          const uint64_t x = first ^ codes[cand];
          const int pfx = x ? __builtin_clzll(x) : 64;
          if (pfx > prefix) at = cand;
        }
      } while (step > 1);
      const size_t s = at + 1;
      const size_t big = std::max(s - lo, hi - s);
      if (ceilLog2(big) <= kStackDepth - 1 - depth) split = s;  // keep internal depth <= kStackDepth-1
    }
    int32_t h0, h1;
    const int32_t c0 = buildRange(lo, split, depth + 1, h0);
    const int32_t c1 = buildRange(split, hi, depth + 1, h1);
    child0[me] = c0; child1[me] = c1;
    height[me] = 1 + std::max(h0, h1);
    outHeight = height[me];
    return me;
  }

  void build(const ExaBrickRegion *regions, size_t n)
  {
    std::vector<float> boxes(6 * n);
    for (size_t i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) { boxes[6 * i + k] = regions[i].domain_lo[k]; boxes[6 * i + 3 + k] = regions[i].domain_hi[k]; }
    build(boxes.data(), n);
  }

  // boxes: 6 floats (lo, hi) per primitive
  void build(const float *boxes, size_t n)
  {
    float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (size_t i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) {
        lo[k] = std::fmin(lo[k], boxes[6 * i + k]);
        hi[k] = std::fmax(hi[k], boxes[6 * i + 3 + k]);
      }
    std::vector<std::pair<uint64_t, uint32_t>> keyed(n);
    for (size_t i = 0; i < n; i++) {
      uint64_t code = 0;
      for (int k = 0; k < 3; k++) {
        const double c = 0.5 * (double(boxes[6 * i + k]) + double(boxes[6 * i + 3 + k]));
        const double ext = double(hi[k]) - double(lo[k]);
        double u = ext > 0 ? (c - lo[k]) / ext : 0.0;
        u = std::min(std::max(u, 0.0), 1.0);
        const uint64_t q = std::min<uint64_t>(uint64_t(u * 2097152.0), 2097151ull);
        code |= spread21(q) << k;
      }
      keyed[i] = { code, uint32_t(i) };
    }
    std::sort(keyed.begin(), keyed.end());
    codes.resize(n); order.resize(n);
    for (size_t i = 0; i < n; i++) { codes[i] = keyed[i].first; order[i] = keyed[i].second; }
    child0.clear(); child1.clear(); height.clear();
    if (n == 0) return;
    if (n == 1) {                      // one region: a root with one real and one padding child
      child0.push_back(~int32_t(0)); child1.push_back(INT32_MIN); height.push_back(1);
      return;
    }
    child0.reserve(n); child1.reserve(n); height.reserve(n);
    int32_t h;
    buildRange(0, n, 0, h);
  }
};

// The ABI calls run on the handle's device and leave the caller's current device as they found it.
struct DeviceGuard {
  int prev = -1;
  hipError_t err;
  explicit DeviceGuard(int device)
  {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    err = hipSetDevice(device);
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define EXA_ON_DEVICE(h) DeviceGuard guard_((h)->device); HIP_TRY(h, guard_.err)

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count)
  {
    release();
    n = count;
    if (count == 0) return hipSuccess;
    // 16 spare bytes: the pair load of a one-cell-wide row reads one float past the last brick
    return hipMalloc((void **)&p, count * sizeof(T) + 16);
  }
  hipError_t upload(const T *src, size_t count)
  {
    hipError_t e = alloc(count);
    if (e != hipSuccess || count == 0) return e;
    return hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
  }
  // like upload, but keeps the allocation when it is large enough (per-frame tables)
  hipError_t refill(const T *src, size_t count)
  {
    if (count > cap || !p) {
      hipError_t e = alloc(std::max(count, size_t(1)));
      if (e != hipSuccess) return e;
      cap = std::max(count, size_t(1));
    }
    n = count;
    return count ? hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice) : hipSuccess;
  }
  size_t cap = 0;
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; cap = 0; }
  ~DevBuf() { release(); }
};

} // namespace

struct ExaHipRenderer {
  int device = 0;
  std::string err;
  void fail(const std::string &m) { err = m; }

  // multi-device handle (exa_hip_create_multi): this object only fans out to `children`, one complete renderer per
  // entry of the device list, each owning the tiles t with t % n == i and storing them straight into the root
  // device's row-major frame (peer-mapped pointer)
  std::vector<ExaHipRenderer *> children;
  bool colorRowMajor = false;          // a child: colour goes row-major into the destination frame
  hipStream_t ownStream = nullptr;     // a child's launch stream
  hipEvent_t evCall = nullptr;         // multi handle: the caller's stream position at the start of a frame

  // scene
  DevBuf<int4> bricks;
  DevBuf<int32_t> leafList;
  DevBuf<int4> leafHdr;
  DevBuf<float> scalars;
  // channel-interleaved copy of the primary channels, float[cell][ilChannels], for the multi-channel march (built on the
  // device by the first frame that marches 2..4 channels; the field-major arrays of the ABI stay for everything else)
  DevBuf<float> cellsIl;
  int ilChannels = 0;
  int ilNoMemory = 0;                // channel count whose interleaved copy could not be allocated (not tried again)
  int interleave = 1;                // option "interleave"
 bool emptyCells = false;           // the scene is marked allowEmptyCells (the reference's ALLOW_EMPTY_CELLS build): source-order kernels with the poison test
  int basisForm = 1;                 // option "basis_form": 1 (default) = the eight-corner basis sums per axis with fused multiply-adds, 0 = in the reference's source order
  int addr64 = 0;                    // option "addr64": the general 64-bit address form even where 32-bit offsets would do (tests)
  int packRecords = 1;               // option "pack_records": 0 = the march takes region ids and loads the region records, as in scenes
                                     // whose records {first brick, brick count, level} do not fit the 32 bits of a leaf reference (tests)
  uint64_t totalCells = 0;
  // Order of the bricks' cells in memory (option brick_order): 0 = as uploaded (the running `begin` of
  // OptixRenderer.cpp:71-93), 1 = along a Morton curve of the brick centres.  Cells are only ever found through their
  // brick's `begin`, so the module may move them; switching re-lays the fields on the device.
  std::vector<uint32_t> beginUploaded, beginMorton;   // per brick
  int brickOrder = 0, brickOrderWanted = 0;
  bool brickOrderPossible = true;   // the scene has the reference's layout (fields at f * totalCells, begins a partition): cells may be moved
  uint64_t numBricks = 0, leafListSize = 0;
  int applyBrickOrder(hipStream_t s)
  {
    if (brickOrderWanted == brickOrder) return 0;
    const std::vector<uint32_t> &from = brickOrder ? beginMorton : beginUploaded, &to = brickOrderWanted ? beginMorton : beginUploaded;
    HIP_TRY(this, hipStreamSynchronize(s));
    HIP_TRY(this, hipDeviceSynchronize());                 // frames in flight on other streams read the old layout
    DevBuf<uint32_t> dFrom, dTo;
    DevBuf<float> moved;
    HIP_TRY(this, dFrom.upload(from.data(), from.size()));
    HIP_TRY(this, dTo.upload(to.data(), to.size()));
    HIP_TRY(this, moved.alloc(scalars.n));
    HIP_TRY(this, launchPermuteBricks(scalars.p, moved.p, dFrom.p, dTo.p, bricks.p, numBricks, leafHdr.p, leafList.p, leafListSize,
                                      totalCells, numFields, s));
    HIP_TRY(this, hipStreamSynchronize(s));
    std::swap(scalars.p, moved.p);                         // `moved` now owns the old array and frees it
    sc.scalars = scalars.p;
    brickOrder = brickOrderWanted;
    ilChannels = 0; cellsIl.release();                     // the interleaved copy follows the new order
    return 0;
  }
  DevBuf<RegionInfo> regionInfo;
  DevBuf<float2> valueRange;
  DevBuf<float> domain;
  DeviceScene sc{};
  int numFields = 0;

  // region kd-tree (optional; exact front-to-back walk)
  DevBuf<KdNodeDev> kdNodes;
  DevBuf<KdNodeDev> kdMarchNodes;       // copy of kdNodes whose leaf references are packed region records (may be empty)
  int32_t kdMarchRoot = 0;
  uint32_t leafBeginBits = 0, leafSizeBits = 0;
  DevBuf<RegionRec> regionRec;
  DevBuf<int32_t> kdLevelIds;
  std::vector<int> kdLevelBegin;
  int32_t kdRoot = EXA_KD_EMPTY;
  bool rootLeafVolActive = true, rootLeafIsoActive = true;   // activity of the only region when the kd tree is one leaf
  bool haveKd = false;
  int accel = 1;                     // 1 = kd walk when available, 0 = LBVH
  float kdLo[3], kdHi[3];

  // Rope walk of the DVR march (option "walk": 0 = chosen per frame, 1 = the stack walk, 2 = the rope walk).  The leaves of
  // the kd-tree with their boxes and neighbour links are built on the host at the first frame that wants them (buildRopes);
  // the stack walk skips inactive subtrees, the rope walk passes through every leaf on the ray, so the automatic choice
  // takes the rope walk when at least kRopeActiveFraction of the regions are active for the volume march.
  DevBuf<RopeLeaf> ropeLeaves;
  DevBuf<KdNodeDev> ropeNodes;
  DevBuf<uint32_t> activeCountBuf;
  int32_t ropeRoot = EXA_KD_EMPTY + 1;
  bool ropeBuilt = false, ropeFailed = false, ropeFlagsStale = true, ropeThisFrame = false;
  int ropeFastDiv = 0, ropeAddr32 = 0;
  int walkMode = 0;
  uint32_t activeRegions = 0;        // regions active for the volume march (refreshed with the activity)
  // (C4 scene, kernel ms stack / rope by active fraction: 0.16 3.82 / 5.35, 0.23 5.45 / 7.22, 0.33 6.96 / 7.88, 0.50 1.91 / 1.65,
  //  0.62 1.98 / 1.66, 0.79 2.01 / 1.68, 1.0 19.84 / 17.31; profiles/r05_experiments.txt 5)
  static constexpr double kRopeActiveFraction = 0.4;
  bool ropeWanted() const
  {
    if (!useKd() || ropeFailed || walkMode == 1) return false;
    if (walkMode == 2) return true;
    return double(activeRegions) >= kRopeActiveFraction * double(sc.numRegions);
  }
  int buildRopes();

  // triangle surfaces
  DevBuf<BvhNode> meshNodes;
  DevBuf<float> meshVerts;
  DevBuf<int32_t> meshTris;
  int numTris = 0;

  // streamline tracer
  ExaHipTracer tracer{};
  bool haveTracer = false;
  DevBuf<float> traces;
  DevBuf<BvhNode> streamNodes;
  int numStreamPrims = 0, timestep = 0;
  bool streamDirty = false;

  // LBVH
  DevBuf<BvhNode> volNodes, isoNodes;
  DevBuf<int32_t> levelIds;
  DevBuf<uint8_t> volActive, isoActive;
  bool volDirty = true, isoDirty = true;

  // state
  DevBuf<float4> xf;
  float xfHost[EXA_MAX_CHANNELS][EXA_NUM_XF_VALUES][4];
  bool xfDirty = true;
  ExaHipFrameState fs{};
  ExaHipParams p{};
  bool haveFs = false, haveParams = false;

  // framebuffer / shard
  int W = 0, H = 0, tilesX = 0, tilesY = 0;
  int rank = 0, world = 1;
  int tileOrder = 4;                 // Z-order launch sequence (measured best on C4)
  int debugPixel = -1;
  int fastMath = 1;                  // hardware exp2/log2 for the opacity correction (kd kernel)
  int mul24 = 0, addr32 = 0;         // address arithmetic the scene's sizes allow (set at creation)
  int fastSampler = 1;               // option fast_sampler (surfaces pre-pass; 0 = the literal addBasisFunctions)
  int tfFilter = 1;                  // TF filter weight in 1.8 fixed point as CUDA's tex1D (0: full precision)
  float tfFracMagic() const { return tfFilter ? 32768.f : 0.f; }
  DevBuf<float4> accum;
  DevBuf<float4> surf;
  DevBuf<uint32_t> tileCost;            // launch-order feedback, one entry per tile of the image
  // Launch plan of a frame with surfaces (option prepass_split, default 1).  The surfaces pre-pass is bound by the LATENCY
  // of its longest iso marches (C3: 0.3 G vector instructions in 3.6 ms), the march behind it by throughput.  The frame
  // that measures tile costs also records every tile's longest iso march; afterwards the few tiles with long pre-pass
  // rays ("heavy") get their own pre-pass + march pipeline on a side stream, which runs beside the pre-pass + march of
  // all other tiles instead of in front of it.  Same launches per tile, same pixels.
  DevBuf<uint32_t> tileCostPre;
  DevBuf<int32_t> splitMap;             // cheap tiles in launch order, then the heavy ones
  int nPreCheap = 0, nPreHeavy = 0;
  int prepassSplit = 1;
  bool preMeasured = false;             // the cost frame had surfaces (tileCostPre is valid)
  std::vector<int32_t> baseMap, curMap; // static launch order (tile_order) / the order in use
  int feedback = 1;                     // option tile_feedback
  int statsMode = 1;                    // option stats_mode: what exa_hip_render_stats collects (1 work counters, 2 wave time by phase)
  int costPhase = 0;                    // 1: the next synchronous frame measures tile costs, then the tiles are re-ordered
  // wide march (L lanes per ray) for the tiles on the frame's critical path
  int wideMode = 1;                     // option wide_march: 0 off, 1 by cost, 2 / 4 every tile with that many lanes (tests)
  // listed leaves per ray of a wide tile: kWideSegCap per window
  static size_t segsPerRay(int lanes) { return size_t(lanes) * kWideSegCap; }
  int numSimdWaves = 256 * 4 * 6;       // waves the device holds at the march kernel's occupancy
  DevBuf<int32_t> normalMap, wideMap;   // one-lane tiles in launch order; wide tiles, the 4-lane ones first
  DevBuf<float4> wideSegs;              // leaf lists of the wide march's window walkers (grown on demand)
  int nNormal = 0, nWide4 = 0, nWide2 = 0;
  int lanesTopInUse = 4;                // lanes per ray of the nWide4 tiles of the current plan
  hipStream_t side4 = nullptr, side2 = nullptr, sideN = nullptr;
  hipEvent_t evFork = nullptr, evJoin4 = nullptr, evJoin2 = nullptr, evJoinN = nullptr;
  DevBuf<uint32_t> surfRnd;
  // option ao_overlap (default 0): 1 = the deferred AO rays run BESIDE the march instead of in front of it.  The march needs the
  // surfaces' hit distance up front but their colour only for its very last operation, and the AO launch — as long as its
  // longest rays, with few waves busy — writes nothing but that colour: the march stores its pixel colour (pixBuf) and a small
  // kernel finishes the pixels once both are done.  Same operations per pixel, same order.  Measured on C5: -0.6 % beside the
  // six-wave march (1079.9 -> 1073.5 ms), +1.4 % beside the seven-wave march with four frames in flight (1052 -> 1067 ms; a
  // lone frame: 1049 / 1050) — the march now fills the GPU on its own and the finishing pass is extra traffic —, hence off.
  DevBuf<float4> pixBuf;
  int aoOverlap = 0;
  hipEvent_t evPre = nullptr, evPre2 = nullptr, evAo = nullptr, evAo2 = nullptr;
  DevBuf<AoRecord> aoRecs;              // deferred AO rays: one record per shaded hit and pixel slot at most
  DevBuf<uint32_t> aoCount;             // [0..3] the frame's list (or the cheap pipeline's), [4..7] the heavy pipeline's
  DevBuf<uint32_t> aoKeys, aoOrder, aoHist;   // ao_defer = 2: bin of every listed ray, ray indices in bin order, 2 x aoBins counters (one set per pipeline)
  DevBuf<uint8_t> aoHit;                // ... and the rays' hit flags
  uint32_t aoBins = 0;
  int aoDefer = 1;                      // option ao_defer: 1 (default since round 4: C5 1317 vs 1329 ms per 16-sample frame, C3 + iso + AO 15.8 vs 16.4 ms), 0 inline, 2 sorted
  DevBuf<uint32_t> color;
  DevBuf<int32_t> tileMap;
  int numBlocks = 0;
  bool layoutDirty = true;

  DevBuf<unsigned long long> statsBuf;
  int walkProbeOn = 0;                  // option walk_probe
  DevBuf<uint32_t> walkProbe;
  DevBuf<int32_t> errorFlag;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
  ExaHipStats last{};

  bool isoEnabled() const
  {
    for (int i = 0; i < EXA_MAX_ISO_SURFACES; i++) if (fs.iso[i].enabled) return true;
    return false;
  }
  bool contourEnabled() const
  {
    for (int i = 0; i < EXA_MAX_CONTOUR_PLANES; i++) if (fs.contour[i].enabled) return true;
    return false;
  }
  bool surfacesEnabled() const { return isoEnabled() || contourEnabled() || numTris > 0 || numStreamPrims > 0; }
  float voxLo[3], voxHi[3];
  // worldSpaceBounds = rcp(voxelSpaceTransform) applied to the voxel bounds (OptixRenderer.cpp:330-332);
  // rcp(affine3f) = inverse of the linear part by adjoint/determinant, p' = -(L^-1 p)
  void worldBounds(float lo[3], float hi[3]) const
  {
    auto cross = [](const float *a, const float *b, float *o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; };
    float c0[3], c1[3], c2[3];
    cross(fs.xfm_vy, fs.xfm_vz, c0); cross(fs.xfm_vz, fs.xfm_vx, c1); cross(fs.xfm_vx, fs.xfm_vy, c2);
    const float det = fs.xfm_vx[0] * c0[0] + fs.xfm_vx[1] * c0[1] + fs.xfm_vx[2] * c0[2];
    const float ix[3] = { c0[0] / det, c1[0] / det, c2[0] / det }, iy[3] = { c0[1] / det, c1[1] / det, c2[1] / det },
                iz[3] = { c0[2] / det, c1[2] / det, c2[2] / det };
    float ip[3];
    for (int k = 0; k < 3; k++) ip[k] = -(fs.xfm_p[0] * ix[k] + (fs.xfm_p[1] * iy[k] + fs.xfm_p[2] * iz[k]));
    for (int k = 0; k < 3; k++) {
      lo[k] = voxLo[0] * ix[k] + (voxLo[1] * iy[k] + (voxLo[2] * iz[k] + ip[k]));
      hi[k] = voxHi[0] * ix[k] + (voxHi[1] * iy[k] + (voxHi[2] * iz[k] + ip[k]));
    }
  }
  uint64_t outputPixels() const { return uint64_t(numBlocksFor()) * kTilePixels; }
  int numBlocksFor() const
  {
    const int tiles = tilesX * tilesY;
    if (world <= 1) return tiles;
    return tiles > rank ? (tiles - rank + world - 1) / world : 0;
  }

  int rebuildLayout()
  {
    tilesX = (W + kTile - 1) / kTile;
    tilesY = (H + kTile - 1) / kTile;
    numBlocks = numBlocksFor();
    const size_t px = world <= 1 ? size_t(W) * H : size_t(numBlocks) * kTilePixels;
    HIP_TRY(this, accum.alloc(px));
    if (px) HIP_TRY(this, hipMemset(accum.p, 0, px * sizeof(float4)));
    HIP_TRY(this, color.alloc(px));
    surf.release(); surfRnd.release();       // allocated by the first frame that has surfaces
    std::vector<int32_t> map;
    map.reserve(numBlocks);
    for (int t = rank; t < tilesX * tilesY; t += world) map.push_back(t);
    if (tileOrder == 1 && world == 1 && tilesX % 8 == 0 && tilesY % 8 == 0 && ((tilesX / 8) * (tilesY / 8)) % 8 == 0) {
      // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so block b
      // lands on XCD b%8.  Give each XCD whole 8x8-tile supertiles (128x128 px) so the
      // rays sharing bricks also share one L2.
      const int stx = tilesX / 8;
      for (int b = 0; b < numBlocks; b++) {
        const int xcd = b % 8, j = b / 8;
        const int super = (j / 64) * 8 + xcd, in = j % 64;
        const int sx = super % stx, sy = super / stx;
        map[b] = (sy * 8 + in / 8) * tilesX + sx * 8 + in % 8;
      }
    }
    if (tileOrder == 2) {          // fixed pseudo-random permutation (load-balance experiment)
      uint64_t st = 0x9E3779B97F4A7C15ull;
      for (size_t i = map.size(); i > 1; i--) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        std::swap(map[i - 1], map[size_t((st >> 33) % i)]);
      }
    } else if (tileOrder == 3) {   // centre-out: tiles nearest the image centre first
      const float cx = 0.5f * tilesX, cy = 0.5f * tilesY;
      std::stable_sort(map.begin(), map.end(), [&](int32_t a, int32_t b) {
        const float ax = a % tilesX + 0.5f - cx, ay = a / tilesX + 0.5f - cy, bx = b % tilesX + 0.5f - cx, by = b / tilesX + 0.5f - cy;
        return ax * ax + ay * ay < bx * bx + by * by;
      });
    }
    if (tileOrder >= 4) {          // Z-order: tiles in flight form a compact 2-d patch of the image
      auto part = [](uint32_t v) { v &= 0xffff; v = (v | v << 8) & 0x00ff00ff; v = (v | v << 4) & 0x0f0f0f0f; v = (v | v << 2) & 0x33333333; v = (v | v << 1) & 0x55555555; return v; };
      std::stable_sort(map.begin(), map.end(), [&](int32_t a, int32_t b) {
        return (part(a % tilesX) | part(a / tilesX) << 1) < (part(b % tilesX) | part(b / tilesX) << 1);
      });
      // 5..7: deal chunks of 16/64/256 Z-consecutive tiles to the 8 XCDs (block b runs on XCD b%8)
      const int chunk = tileOrder == 5 ? 16 : (tileOrder == 6 ? 64 : (tileOrder == 7 ? 256 : 0));
      if (chunk && map.size() % size_t(8 * chunk) == 0) {
        std::vector<int32_t> z(map);
        for (size_t b = 0; b < map.size(); b++) {
          const size_t xcd = b % 8, j = b / 8;
          map[b] = z[((j / chunk) * 8 + xcd) * chunk + j % chunk];
        }
      }
    }
    HIP_TRY(this, tileMap.upload(map.data(), map.size()));
    HIP_TRY(this, tileCost.alloc(size_t(tilesX) * tilesY));
    HIP_TRY(this, tileCostPre.alloc(size_t(tilesX) * tilesY));
    nPreCheap = nPreHeavy = 0;
    baseMap = map; curMap = map;
    costPhase = 1;
    nNormal = nWide4 = nWide2 = 0;
    if (assignWide(nullptr)) return 1;
    layoutDirty = false;
    return 0;
  }

  // Launch-order feedback.  A frame's critical path is its longest rays (a wave runs until its
  // slowest lane is done); launched late they drain alone on an empty GPU — worst on a multi-GPU
  // shard, where a rank holds little more than one GPU-full of waves.  The frame after a change of
  // view/TF/layout records each tile's longest wave (march iterations); from then on the heaviest
  // tiles are launched first (coarse cost classes, the static order inside a class so that the
  // tiles in flight still share bricks).  Pixels do not depend on the launch order.
  int reorderFromCosts()
  {
    costPhase = 0;
    const size_t n = curMap.size();
    if (n < 2) return 0;
    std::vector<uint32_t> costOfTile(size_t(tilesX) * tilesY, 0);
    HIP_TRY(this, hipMemcpy(costOfTile.data(), tileCost.p, costOfTile.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    uint32_t maxC = 0;
    for (size_t b = 0; b < n; b++) maxC = std::max(maxC, costOfTile[curMap[b]]);
    // cost classes: heaviest first between classes, the static (Z-order) sequence inside a class, so that the tiles in
    // flight still share bricks.  Measured on C4 (EXA_COST_CLASSES = 1 / 2 / 4 / 8 / 16 / 32 / 64 / 256 / 1024 / 4096):
    // 24.29 / 24.11 / 23.36 / 22.76 / 22.56 / 22.37 / 22.33 / 22.26 / 22.30 / 22.24 ms — the frame's tail matters more than
    // the locality of the tiles in flight
    int kClasses = 256;
    if (const char *e = std::getenv("EXA_COST_CLASSES")) kClasses = std::max(1, std::min(4096, std::atoi(e)));
    std::vector<std::vector<int32_t>> cls(kClasses);
    for (size_t b = 0; b < n; b++) {
      const int32_t t = baseMap[b];
      const int c = kClasses - 1 - int(uint64_t(costOfTile[t]) * kClasses / (uint64_t(maxC) + 1));
      cls[c].push_back(t);
    }
    if (std::getenv("EXA_HIP_VERBOSE")) {
      uint64_t sum = 0;
      for (size_t b = 0; b < n; b++) sum += costOfTile[curMap[b]];
      std::fprintf(stderr, "[exa_hip] tile costs: %zu tiles, max %u iterations, sum %llu, per class (heaviest first):", n, maxC,
                   (unsigned long long)sum);
      for (int c = 0; c < kClasses; c++) std::fprintf(stderr, " %zu", cls[c].size());
      std::fprintf(stderr, "\n");
    }
    std::vector<int32_t> order;
    order.reserve(n);
    for (int c = 0; c < kClasses; c++) order.insert(order.end(), cls[c].begin(), cls[c].end());
    if (order != curMap) {
      HIP_TRY(this, hipMemcpy(tileMap.p, order.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
      curMap.swap(order);
    }
    nPreCheap = nPreHeavy = 0;
    if (preMeasured && prepassSplit) {
      std::vector<uint32_t> pre(size_t(tilesX) * tilesY, 0);
      HIP_TRY(this, hipMemcpy(pre.data(), tileCostPre.p, pre.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
      uint32_t maxP = 0;
      for (size_t b = 0; b < n; b++) maxP = std::max(maxP, pre[curMap[b]]);
      // heavy: a longest iso march above 1/32 of the frame's longest (and long enough to matter at all)
      // (C3 / C5 with the threshold at max/8 | >= 64 steps: 16.81 / 1444 ms, max/32 | 16: 16.61 / 1444, every tile with any iso
      // step: 16.59 / 1444 — on C5 only 9 % of the tiles have any; profiles/r03_prepass_split_threshold.txt)
      uint32_t div = 32, minSteps = 16;
      if (const char *e = std::getenv("EXA_PREPASS_SPLIT_DIV")) div = (uint32_t)std::max(1, std::atoi(e));          // calibration runs
      if (const char *e = std::getenv("EXA_PREPASS_SPLIT_MIN")) minSteps = (uint32_t)std::max(0, std::atoi(e));
      const uint32_t thr = std::max<uint32_t>(minSteps, maxP / div);
      std::vector<int32_t> cheap, heavy;
      for (size_t b = 0; b < n; b++) (pre[curMap[b]] > thr ? heavy : cheap).push_back(curMap[b]);
      if (!heavy.empty() && !cheap.empty()) {
        // the heavy pipeline starts with its longest pre-pass rays
        std::stable_sort(heavy.begin(), heavy.end(), [&](int32_t x, int32_t y) { return pre[x] > pre[y]; });
        std::vector<int32_t> both(cheap);
        both.insert(both.end(), heavy.begin(), heavy.end());
        HIP_TRY(this, splitMap.refill(both.data(), both.size()));
        nPreCheap = (int)cheap.size(); nPreHeavy = (int)heavy.size();
      }
      if (std::getenv("EXA_HIP_VERBOSE"))
        std::fprintf(stderr, "[exa_hip] pre-pass costs: longest iso march %u steps; %d tiles in the heavy pipeline, %d in the other\n", maxP, nPreHeavy, nPreCheap);
    }
    return assignWide(&costOfTile);
  }

  // Which tiles march with 2 or 4 lanes per ray.  In units of one wave's march iterations on an idle
  // GPU: a tile's critical path is cost / speedup(L); the time the GPU needs for everything else is
  // the work (4 waves per tile, x L / speedup(L) for a wide tile) over the waves it holds, at the pace
  // of a loaded GPU.  Heaviest tiles first, each gets the smallest L that brings its path below the
  // fill time; on one GPU nothing qualifies, on a shard of 8 the few hundred longest tiles do.
  int assignWide(const std::vector<uint32_t> *costOfTile)
  {
    const size_t n = curMap.size();
    std::vector<int32_t> normal, w4, w2;
    const int lanesTop = 4;
    if (wideMode == 2 || wideMode == 4) {
      (wideMode >= 4 ? w4 : w2) = curMap;
    } else if (wideMode == 1 && costOfTile) {
      // Model constants: a critical tile finishes kSpeed2 (2 lanes) times sooner and costs kWork2 / kWork4
      // times the work; a loaded GPU steps a wave 1.3x slower.  Measured with the round-1 kernels (probe 6.0 -> 3.7 ->
      // 2.5 ms).  With the round-2 kernels the probe (tests/gpu_wide_probe.py) gives 4.78 -> 3.39 -> 2.52 ms (8 lanes:
      // 3.00 ms, not instantiated) and every tile of a rank forced wide costs x1.9 / x2.4 the time (tests/gpu_shard_modes.py);
      // variations of the constants around these values move the shard of 8 by +-0.2 ms (4.46 .. 4.95 ms), the set below
      // stays within 0.05 ms of the best one tried.
      // NOTE: the module owns exactly three side streams.  A fourth (tried for an 8-lane class) made two of the
      // streams that carry one frame's launches share a hardware queue, and the shard of 8 went from 4.6 to 7.0 ms.
      double kSpeed2 = EXA_WIDE_SPEED2;
      const double kWork2 = EXA_WIDE_WORK2, kLoaded = 1.3;
      double kWork4 = EXA_WIDE_WORK4;
      if (const char *e = std::getenv("EXA_WIDE_WORK_TOP")) kWork4 = std::atof(e);          // calibration runs
      if (const char *e = std::getenv("EXA_WIDE_SPEED2")) kSpeed2 = std::atof(e);
      double fill = 0;
      for (size_t b = 0; b < n; b++) fill += 4.0 * (*costOfTile)[curMap[b]];
      fill *= kLoaded / numSimdWaves;
      // curMap is ordered by descending cost class; inside a class the decision only depends on the tile's own cost
      for (size_t b = 0; b < n; b++) {
        const int32_t t = curMap[b];
        const double c = (*costOfTile)[t];
        int L = 1;
        if (c > fill) L = (c / kSpeed2 > fill) ? 4 : 2;
        if (L == 1) { normal.push_back(t); continue; }
        fill += 4.0 * c * ((L == 4 ? kWork4 : kWork2) - 1.0) * kLoaded / numSimdWaves;
        (L == 4 ? w4 : w2).push_back(t);
      }
    } else {
      normal = curMap;
    }
    {
      // leaf lists: 16 B x kWideSegCap per window walker = 8 KiB per lane; keep them within 8 GiB by handing the
      // lightest wide tiles back to the one-lane march (forced modes on large frames)
      const size_t perTile4 = size_t(kTilePixels) * segsPerRay(lanesTop) * sizeof(float4), perTile2 = size_t(kTilePixels) * segsPerRay(2) * sizeof(float4);
      const size_t budget = size_t(std::getenv("EXA_WIDE_BUDGET_GB") ? std::atoi(std::getenv("EXA_WIDE_BUDGET_GB")) : 8) << 30;
      while (w4.size() * perTile4 + w2.size() * perTile2 > budget) {
        if (!w2.empty()) { normal.push_back(w2.back()); w2.pop_back(); }
        else { normal.push_back(w4.back()); w4.pop_back(); }
      }
    }
    if (w4.empty() && w2.empty()) { nNormal = (int)n; nWide4 = nWide2 = 0; return 0; }
    std::vector<int32_t> wide(w4);
    wide.insert(wide.end(), w2.begin(), w2.end());
    {
      const size_t need = (w4.size() * segsPerRay(lanesTop) + w2.size() * segsPerRay(2)) * size_t(kTilePixels);
      if (need > wideSegs.n && wideSegs.alloc(need) != hipSuccess) {
        // no room for the leaf lists: the frame simply keeps the one-lane march
        (void)hipGetLastError();
        wideSegs.release();
        nNormal = (int)n; nWide4 = nWide2 = 0;
        return 0;
      }
    }
    HIP_TRY(this, normalMap.refill(normal.data(), normal.size()));
    HIP_TRY(this, wideMap.refill(wide.data(), wide.size()));
    nNormal = (int)normal.size(); nWide4 = (int)w4.size(); nWide2 = (int)w2.size();
    lanesTopInUse = lanesTop;
    if (std::getenv("EXA_HIP_VERBOSE"))
      std::fprintf(stderr, "[exa_hip] wide march: %d tiles x%d lanes, %d x2, %d one lane per ray\n", nWide4, lanesTop, nWide2, nNormal);
    return 0;
  }

  int kdRefit(const uint8_t *active, int which, hipStream_t s)
  {
    for (size_t h = 0; h + 1 < kdLevelBegin.size(); h++)
      HIP_TRY(this, launchKdRefit(kdNodes.p, kdMarchNodes.p, kdLevelIds.p + kdLevelBegin[h], kdLevelBegin[h + 1] - kdLevelBegin[h],
                                  active, which, s));
    return 0;
  }
  bool useKd() const { return haveKd && accel == 1; }

  // The LBVH over the regions (north_star's structure; accel=0, scenes without a kd-tree, and the
  // streamline tracer's point queries) is built on first use, on the device (exa_lbvh.hip): Morton codes, radix
  // sort, topology level by level; boxes are filled by the refit.  Option lbvh_build = 1 builds the same tree on
  // the host instead (LbvhTopology above; the two are identical node for node, tests compare them).
  bool lbvhBuilt = false;
  int lbvhOnHost = 0;
  std::vector<std::pair<int, int>> levelRanges;   // (offset into levelIds, count) per refit launch, children before parents
  DevBuf<BvhNode> topoNodes;                       // children filled, boxes empty: the template of volNodes / isoNodes
  int ensureLbvh()
  {
    if (lbvhBuilt) return 0;
    const size_t nr = domain.n / 6;
    levelRanges.clear();
    const auto tBuild0 = std::chrono::steady_clock::now();
    if (nr < 2 || lbvhOnHost) {
      std::vector<float> boxes(domain.n);
      HIP_TRY(this, hipMemcpy(boxes.data(), domain.p, domain.n * sizeof(float), hipMemcpyDeviceToHost));
      LbvhTopology topo;
      topo.build(boxes.data(), nr);
      boxes.clear(); boxes.shrink_to_fit();
      const size_t ni = topo.child0.size();
      std::vector<BvhNode> tmpl(ni);
      for (size_t i = 0; i < ni; i++) {
        BvhNode &n = tmpl[i];
        n.q0 = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX);
        n.q1 = make_float4(-FLT_MAX, -FLT_MAX, FLT_MAX, FLT_MAX);
        n.q2 = make_float4(FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
        n.child0 = topo.child0[i]; n.child1 = topo.child1[i]; n.pad0 = n.pad1 = 0;
      }
      HIP_TRY(this, topoNodes.upload(tmpl.data(), ni));
      int maxH = 0;
      for (size_t i = 0; i < ni; i++) maxH = std::max(maxH, topo.height[i]);
      std::vector<int> count(maxH + 2, 0);
      for (size_t i = 0; i < ni; i++) count[topo.height[i]]++;
      std::vector<int> begin(1, 0);
      for (int hh = 1; hh <= maxH; hh++) begin.push_back(begin.back() + count[hh]);
      std::vector<int32_t> ids(ni);
      {
        std::vector<int> cursor(begin.begin(), begin.end());
        for (size_t i = 0; i < ni; i++) ids[cursor[topo.height[i] - 1]++] = (int32_t)i;
      }
      for (int hh = 1; hh <= maxH; hh++) levelRanges.push_back({ begin[hh - 1], count[hh] });   // by height, leaves' parents first
      HIP_TRY(this, levelIds.upload(ids.data(), ids.size()));
      sc.numInternal = (uint32_t)ni;
    } else {
      const size_t ni = nr - 1;
      HIP_TRY(this, topoNodes.alloc(ni));
      HIP_TRY(this, levelIds.alloc(ni));
      std::vector<uint32_t> perDepth;
      HIP_TRY(this, buildLbvhTopologyDevice(domain.p, (uint32_t)nr, topoNodes.p, levelIds.p, perDepth, nullptr));
      int at = 0;
      std::vector<std::pair<int, int>> byDepth;
      for (uint32_t c : perDepth) { byDepth.push_back({ at, (int)c }); at += (int)c; }
      levelRanges.assign(byDepth.rbegin(), byDepth.rend());            // deepest level first
      sc.numInternal = (uint32_t)ni;
    }
    HIP_TRY(this, volNodes.alloc(topoNodes.n));
    HIP_TRY(this, hipMemcpy(volNodes.p, topoNodes.p, topoNodes.n * sizeof(BvhNode), hipMemcpyDeviceToDevice));
    lbvhBuilt = true;
    volDirty = isoDirty = true;          // boxes of both LBVHs come from the next refit
    if (std::getenv("EXA_HIP_VERBOSE")) {
      (void)hipDeviceSynchronize();
      std::fprintf(stderr, "[exa_hip] LBVH over %zu regions built on the %s in %.1f ms (%zu refit launches)\n", nr,
                   (nr < 2 || lbvhOnHost) ? "host" : "device",
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tBuild0).count(), levelRanges.size());
    }
    return 0;
  }
  bool needLbvh() const { return !useKd() || (haveTracer && tracer.enabled); }

  int refit(DevBuf<BvhNode> &nodes, const uint8_t *active, hipStream_t s)
  {
    for (const auto &r : levelRanges)
      HIP_TRY(this, launchRefit(nodes.p, levelIds.p + r.first, r.second, domain.p, active, s));
    return 0;
  }

  int prepareFrame(hipStream_t s)
  {
    if (!haveFs || !haveParams) { fail("exa_hip_render: frame state / params not set"); return 1; }
    if (W <= 0 || H <= 0) { fail("exa_hip_render: framebuffer not sized"); return 1; }
    if (p.numPrimaryChannels < 1 || p.numPrimaryChannels > numFields || p.numChannels > numFields
        || p.colormapChannel < 0 || p.colormapChannel >= numFields) {
      fail("exa_hip_render: channel counts exceed the scene's scalar fields"); return 1;
    }
    if (layoutDirty && rebuildLayout()) return 1;
    if (xfDirty) {
      HIP_TRY(this, hipMemcpyAsync(xf.p, xfHost, sizeof(xfHost), hipMemcpyHostToDevice, s));
      xfDirty = false;
    }
    if (needLbvh() && ensureLbvh()) return 1;
    if (applyBrickOrder(s)) return 1;
    {
      int want = (useKd() && interleave && !emptyCells && p.numPrimaryChannels >= 2 && p.numPrimaryChannels <= 4) ? p.numPrimaryChannels : 0;
      if (want == ilNoMemory) want = 0;                    // this many channels did not fit before: field by field
      if (want != ilChannels) {
        HIP_TRY(this, hipStreamSynchronize(s));            // frames in flight may still read the old copy
        cellsIl.release();
        ilChannels = 0;
        if (want && cellsIl.alloc(size_t(totalCells) * want + 2 * size_t(want)) != hipSuccess) {      // a pair load may reach one cell past the end
          // the copy is an optimisation (want x one field of extra memory): without it the march reads the fields one after
          // the other, same pixels
          (void)hipGetLastError();
          cellsIl.release();
          if (std::getenv("EXA_HIP_VERBOSE"))
            std::fprintf(stderr, "[exa_hip] no memory for the channel-interleaved copy of %d fields: field-by-field march\n", want);
          ilNoMemory = want;
          want = 0;
        }
        if (want) {
          HIP_TRY(this, hipMemsetAsync(cellsIl.p + size_t(totalCells) * want, 0, 2 * size_t(want) * sizeof(float), s));
          HIP_TRY(this, launchInterleave(sc, totalCells, want, cellsIl.p, s));
          ilChannels = want;
        }
      }
    }
    const bool needIso = isoEnabled();
    if (volDirty || (needIso && isoDirty)) {
      HIP_TRY(this, hipEventRecord(ev2, s));
      const bool volChanged = volDirty;
      if (volDirty) {                       // needVolumeBVHRebuild (OptixRenderer.cpp:533-537)
        HIP_TRY(this, launchVolumeActivity(sc, fs, p, xf.p, volActive.p, tfFracMagic(), s));
        if (lbvhBuilt && refit(volNodes, volActive.p, s)) return 1;
        if (haveKd && kdRefit(volActive.p, 0, s)) return 1;
        volDirty = false;
      }
      if (needIso && isoDirty) {            // needIsoBVHRebuild (OptixRenderer.cpp:539-543)
        if (lbvhBuilt && !isoNodes.p && topoNodes.n) {
          HIP_TRY(this, isoNodes.alloc(topoNodes.n));
          HIP_TRY(this, hipMemcpyAsync(isoNodes.p, topoNodes.p, topoNodes.n * sizeof(BvhNode), hipMemcpyDeviceToDevice, s));
        }
        HIP_TRY(this, launchIsoActivity(sc, fs, isoActive.p, s));
        if (lbvhBuilt && refit(isoNodes, isoActive.p, s)) return 1;
        if (haveKd && kdRefit(isoActive.p, 1, s)) return 1;
        isoDirty = false;
      }
      if (haveKd && volChanged) {
        // how many regions the volume march finds active: what the automatic choice of the walk looks at
        if (!activeCountBuf.p) HIP_TRY(this, activeCountBuf.alloc(1));
        HIP_TRY(this, hipMemsetAsync(activeCountBuf.p, 0, sizeof(uint32_t), s));
        HIP_TRY(this, launchRopeActivity(ropeBuilt ? ropeLeaves.p : nullptr, sc.numRegions, volActive.p, 0, activeCountBuf.p, s));
        ropeFlagsStale = !ropeBuilt;
      }
      HIP_TRY(this, hipEventRecord(ev1, s));
      HIP_TRY(this, hipEventSynchronize(ev1));
      HIP_TRY(this, hipEventElapsedTime(&last.rebuild_ms, ev2, ev1));
      if (haveKd && volChanged) HIP_TRY(this, hipMemcpy(&activeRegions, activeCountBuf.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
      if (haveKd && kdRoot < 0 && kdRoot != EXA_KD_EMPTY) {        // the kd tree is a single leaf: its activity lives here
        uint8_t f[2] = {1, 1};
        HIP_TRY(this, hipMemcpy(&f[0], volActive.p + ~kdRoot, 1, hipMemcpyDeviceToHost));
        if (needIso) HIP_TRY(this, hipMemcpy(&f[1], isoActive.p + ~kdRoot, 1, hipMemcpyDeviceToHost));
        rootLeafVolActive = f[0] != 0;
        rootLeafIsoActive = f[1] != 0;
      }
    }
    // which walk this frame's DVR march takes
    const bool ropeBefore = ropeThisFrame;
    ropeThisFrame = ropeWanted();
    if (ropeThisFrame != ropeBefore && std::getenv("EXA_HIP_VERBOSE"))
      std::fprintf(stderr, "[exa_hip] %u of %u regions active for the volume march: %s walk (option walk = %d)\n", activeRegions, sc.numRegions,
                   ropeThisFrame ? "rope" : "stack", walkMode);
    if (ropeThisFrame && !ropeBuilt) {
      HIP_TRY(this, hipStreamSynchronize(s));
      if (buildRopes()) return 1;
      ropeThisFrame = ropeBuilt;
    }
    if (ropeThisFrame && ropeFlagsStale) {
      HIP_TRY(this, launchRopeActivity(ropeLeaves.p, sc.numRegions, volActive.p, 0, nullptr, s));
      ropeFlagsStale = false;
    }
    return 0;
  }

  // needStreamlineBVHRebuild (OptixRenderer.cpp:545-549): BVH over the segments the Streamline bounds
  // program leaves visible (exabrick.cu:541-570), built on the host from the current traces
  int rebuildStreamlines(hipStream_t s)
  {
    streamDirty = false;
    numStreamPrims = 0;
    if (!haveTracer) return 0;
    const int NT = tracer.numTimesteps;
    const long long nprims = (long long)tracer.numTraces * (NT - 1);
    // the device copy of the timestep stops at numTimesteps (advanceTracer uploads it only while <= numTimesteps)
    const int timestep = std::min(this->timestep, NT);
    if (timestep < 2 || nprims <= 0) return 0;
    HIP_TRY(this, hipStreamSynchronize(s));
    std::vector<float> host(traces.n);
    HIP_TRY(this, hipMemcpy(host.data(), traces.p, traces.n * sizeof(float), hipMemcpyDeviceToHost));
    std::vector<float> boxes;
    std::vector<int32_t> prim;
    for (long long p = 0; p < nprims; p++) {
      if (int(p % NT) >= timestep - 1) continue;
      const float *pa = &host[3 * p], *pb = &host[3 * (p + 1)];
      if (!(pa[0] < 2e10f && pb[0] < 2e10f)) continue;
      for (int k = 0; k < 3; k++) boxes.push_back(std::fmin(pa[k] - 2.f, pb[k] - 2.f));
      for (int k = 0; k < 3; k++) boxes.push_back(std::fmax(pa[k] + 2.f, pb[k] + 2.f));
      prim.push_back((int32_t)p);
    }
    if (prim.empty()) return 0;
    LbvhTopology topo;
    topo.build(boxes.data(), prim.size());
    const size_t ni = topo.child0.size();
    std::vector<BvhNode> nodes(ni);
    std::vector<float> nlo(3 * ni), nhi(3 * ni);
    auto childBox = [&](int32_t c, float *lo, float *hi) {
      if (c == INT32_MIN) { for (int k = 0; k < 3; k++) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; } return; }
      if (c < 0) { for (int k = 0; k < 3; k++) { lo[k] = boxes[6 * size_t(~c) + k]; hi[k] = boxes[6 * size_t(~c) + 3 + k]; } return; }
      for (int k = 0; k < 3; k++) { lo[k] = nlo[3 * size_t(c) + k]; hi[k] = nhi[3 * size_t(c) + k]; }
    };
    for (size_t i = ni; i-- > 0;) {
      float l0[3], h0[3], l1[3], h1[3];
      childBox(topo.child0[i], l0, h0);
      childBox(topo.child1[i], l1, h1);
      for (int k = 0; k < 3; k++) { nlo[3 * i + k] = std::fmin(l0[k], l1[k]); nhi[3 * i + k] = std::fmax(h0[k], h1[k]); }
      BvhNode &n = nodes[i];
      n.q0 = make_float4(l0[0], l0[1], l0[2], h0[0]);
      n.q1 = make_float4(h0[1], h0[2], l1[0], l1[1]);
      n.q2 = make_float4(l1[2], h1[0], h1[1], h1[2]);
      auto leaf = [&](int32_t c) { return (c < 0 && c != INT32_MIN) ? ~prim[size_t(~c)] : c; };   // leaf = flat segment index
      n.child0 = leaf(topo.child0[i]); n.child1 = leaf(topo.child1[i]); n.pad0 = n.pad1 = 0;
    }
    HIP_TRY(this, streamNodes.upload(nodes.data(), nodes.size()));
    numStreamPrims = (int)prim.size();
    return 0;
  }

  bool measureCosts = false;            // set by renderImpl for synchronous frames
  int launch(uint32_t *dstDevice, bool stats, hipStream_t s)
  {
    if (streamDirty && rebuildStreamlines(s)) return 1;
    RenderArgs a{};
    a.sc = sc;
    a.volNodes = volNodes.p;
    a.isoNodes = isoNodes.p;
    a.fs = fs;
    a.p = p;
    a.xf = xf.p;
    a.tfFracMagic = tfFracMagic();
    a.fastSampler = fastSampler;
    a.mul24 = mul24; a.addr32 = addr64 ? 0 : addr32;
    a.cellsIl = ilChannels == p.numPrimaryChannels ? cellsIl.p : nullptr;
    a.il32 = (uint64_t(totalCells) + 2) * uint64_t(ilChannels > 0 ? ilChannels : 1) * sizeof(float) <= (1ull << 32) && !addr64 ? 1 : 0;
    {
      // launch.dt a power of two (the reference's default 0.5 is): 1/dt is exact and x/dt == x*(1/dt)
      int e = 0;
      const float mant = std::frexp(p.dt, &e);
      a.invDtPow2 = (mant == 0.5f && e > -100 && e < 100) ? 1.f / p.dt : 0.f;
    }
    a.numXfChannels = numFields;
    a.W = W; a.H = H; a.tilesX = tilesX; a.tilesY = tilesY;
    a.rank = rank; a.world = world;
    a.tileMap = tileMap.p;
    a.color = dstDevice;
    a.colorRowMajor = colorRowMajor ? 1 : 0;
    a.accum = accum.p;
    a.surf = surf.p;
    a.surfRnd = surfRnd.p;
    a.stats = statsBuf.p;
    a.errorFlag = errorFlag.p;
    a.debugPixel = debugPixel;
    a.walkProbe = nullptr;
    if (stats && statsMode == 1 && walkProbeOn && useKd()) {
      const size_t need = size_t(numBlocks) * (256 / 64) * kWalkProbeSize;
      if (walkProbe.n != need) HIP_TRY(this, walkProbe.alloc(need));
      HIP_TRY(this, hipMemsetAsync(walkProbe.p, 0, need * sizeof(uint32_t), s));
      a.walkProbe = walkProbe.p;
    }
    a.tileCost = nullptr;
    a.tileCostPre = nullptr;
    if (feedback && costPhase == 1 && useKd() && measureCosts) {
      HIP_TRY(this, hipMemsetAsync(tileCost.p, 0, tileCost.n * sizeof(uint32_t), s));
      a.tileCost = tileCost.p;
      preMeasured = surfacesEnabled();
      if (preMeasured) {
        HIP_TRY(this, hipMemsetAsync(tileCostPre.p, 0, tileCostPre.n * sizeof(uint32_t), s));
        a.tileCostPre = tileCostPre.p;
      }
    }
    a.kdNodes = kdNodes.p;
    // the instrumented counters re-check every leaf against its region record, so they walk the tree with region ids
    const bool packed = packRecords && kdMarchNodes.p != nullptr && !(stats && statsMode == 1);
    a.kdMarchNodes = packed ? kdMarchNodes.p : kdNodes.p;
    a.kdMarchRoot = packed ? kdMarchRoot : kdRoot;
    // A tree that is one leaf (a one-region scene) has no node to carry the activity bits: the walks start at
    // "done" when that region is inactive (the reference's BVHs hold no primitive then)
    if (!rootLeafVolActive) a.kdMarchRoot = EXA_KD_EMPTY + 1;
    a.leafBeginBits = packed ? leafBeginBits : 0;
    a.leafSizeBits = packed ? leafSizeBits : 0;
    a.regionRec = regionRec.p;
    const bool rope = ropeThisFrame && ropeBuilt;
    a.ropeLeaves = rope ? ropeLeaves.p : nullptr;
    a.ropeNodes = ropeNodes.p;
    a.ropeRoot = ropeRoot;
    a.ropeFastDiv = ropeFastDiv;
    a.ropeAddr32 = addr64 ? 0 : ropeAddr32;
    a.kdRoot = kdRoot;
    a.kdIsoRoot = rootLeafIsoActive ? kdRoot : EXA_KD_EMPTY + 1;
    for (int k = 0; k < 3; k++) { a.kdLo[k] = kdLo[k]; a.kdHi[k] = kdHi[k]; }
    worldBounds(a.worldLo, a.worldHi);
    a.meshNodes = meshNodes.p; a.meshVerts = meshVerts.p; a.meshTris = meshTris.p; a.numTris = numTris;
    a.streamNodes = streamNodes.p; a.traces = traces.p; a.numStreamPrims = numStreamPrims;
    for (int k = 0; k < 3; k++) a.tracerChannels[k] = tracer.channels[k];
    a.numTraces = tracer.numTraces; a.numTimesteps = tracer.numTimesteps; a.timestep = std::min(timestep, tracer.numTimesteps); a.steplen = tracer.steplen;
    if (haveTracer && tracer.enabled && timestep < tracer.numTimesteps && timestep >= 1) {
      // computeTraces: the threads with pixelIdx < numTraces (exabrick.cu:1539)
      const long long px = (long long)W * H;
      HIP_TRY(this, EXA_FORM(launchComputeTraces)(a, traces.p, (int)std::min<long long>(tracer.numTraces, px), s));
    }
    if (useKd() && surfacesEnabled() && surf.n != accum.n) {
      HIP_TRY(this, surf.alloc(accum.n));
      HIP_TRY(this, surfRnd.alloc(accum.n));
      a.surf = surf.p; a.surfRnd = surfRnd.p;
    }
    a.aoRecs = nullptr; a.aoCount = nullptr; a.aoKeys = nullptr; a.aoHist = nullptr; a.aoOrder = nullptr; a.aoHit = nullptr; a.aoBins = 0;
    if (useKd() && surfacesEnabled() && fs.ao.enabled && aoDefer && !stats) {
      // one record per pixel of every launched tile, the padding pixels of partial edge tiles included: the heavy
      // pipeline's list starts behind nPreCheap WHOLE tiles (accum.n = W * H on one GPU is smaller when W or H is not a
      // multiple of the tile)
      const size_t recs = size_t(numBlocks) * kTilePixels;
      bool listOk = true;
      if (aoRecs.n != recs && aoRecs.alloc(recs) != hipSuccess) {
        // the list is an optimisation (64 B per pixel): without it the AO rays are traced inline behind each pixel's primary ray
        (void)hipGetLastError();
        aoRecs.release();
        listOk = false;
        if (std::getenv("EXA_HIP_VERBOSE")) std::fprintf(stderr, "[exa_hip] no memory for the list of deferred AO rays (%zu records): traced inline\n", recs);
      }
      if (listOk) {
      if (!aoCount.p) HIP_TRY(this, aoCount.alloc(8));        // per pipeline: [0] listed hits, [2] the AO kernel's chunk counter
      HIP_TRY(this, hipMemsetAsync(aoCount.p, 0, 8 * sizeof(uint32_t), s));
      a.aoRecs = aoRecs.p; a.aoCount = aoCount.p;
      }
      a.aoKeys = nullptr;
      if (listOk && aoDefer == 2) {
        // bins: (32x32-pixel blocks of the frame, or groups of four of this shard's tiles) x 24 direction classes
        const uint32_t cells = world <= 1 ? uint32_t((W + 31) / 32) * uint32_t((H + 31) / 32) : uint32_t((numBlocks + 3) / 4);
        const uint32_t bins = std::max(1u, cells) * 24u;
        if (aoKeys.n != 2 * recs) { HIP_TRY(this, aoKeys.alloc(2 * recs)); HIP_TRY(this, aoOrder.alloc(2 * recs)); HIP_TRY(this, aoHit.alloc(2 * recs)); }
        if (aoBins != bins) { HIP_TRY(this, aoHist.alloc(2 * size_t(bins))); aoBins = bins; }
        a.aoKeys = aoKeys.p; a.aoOrder = aoOrder.p; a.aoHit = aoHit.p; a.aoHist = aoHist.p; a.aoBins = bins;
      }
    }
    HIP_TRY(this, hipEventRecord(ev0, s));
    // the one-lane DVR march on the walk chosen for this frame
    auto march = [&](const RenderArgs &ra, int n, bool surfArg, int statsArg, hipStream_t st) -> hipError_t {
      return rope ? EXA_FORM(launchRenderKdRope)(ra, n, p.gradientShadingDVR != 0, fastMath != 0, surfArg, statsArg, st)
                  : EXA_FORM(launchRenderKd)(ra, n, p.gradientShadingDVR != 0, fastMath != 0, surfArg, statsArg, st);
    };
    if (useKd()) {
      const bool surfOn = surfacesEnabled();
      const bool wide = !stats && !emptyCells && nWide4 + nWide2 > 0 && p.numPrimaryChannels == 1 && a.debugPixel < 0;
      const bool split = surfOn && !stats && !wide && costPhase == 0 && !a.tileCost && nPreHeavy > 0 && nPreCheap > 0
                         && nPreHeavy + nPreCheap == numBlocks && a.debugPixel < 0;
      // the deferred AO rays beside the march (see aoOverlap); the viewer's clock heat map times the march kernel itself and
      // keeps the plain sequence
      const bool overlap = aoOverlap && a.aoRecs && fs.ao.enabled && !wide && !(fs.clockScale > 0.f);
      if (overlap) {
        if (pixBuf.n != accum.n) HIP_TRY(this, pixBuf.alloc(accum.n));
        a.pixOut = pixBuf.p;
      }
      if (split) {
        // two pipelines side by side (see prepassSplit): pre-pass + march of the heavy tiles, pre-pass + march of the rest
        HIP_TRY(this, hipEventRecord(evFork, s));
        RenderArgs ah = a, ac = a;
        ac.tileMap = splitMap.p;
        ah.tileMap = splitMap.p + nPreCheap;
        if (a.aoRecs) {                                   // each pipeline appends to its own list
          ah.aoRecs = a.aoRecs + size_t(nPreCheap) * kTilePixels;
          ah.aoCount = a.aoCount + 4;
          if (a.aoKeys) {
            const size_t off = 2 * size_t(nPreCheap) * kTilePixels;
            ah.aoKeys = a.aoKeys + off; ah.aoOrder = a.aoOrder + off; ah.aoHit = a.aoHit + off; ah.aoHist = a.aoHist + a.aoBins;
          }
        }
        HIP_TRY(this, hipStreamWaitEvent(side2, evFork, 0));
        HIP_TRY(this, EXA_FORM(launchSurfacePrepassKd)(ah, nPreHeavy, false, side2));
        HIP_TRY(this, hipStreamWaitEvent(sideN, evFork, 0));
        HIP_TRY(this, EXA_FORM(launchSurfacePrepassKd)(ac, nPreCheap, false, sideN));
        if (overlap) {
          // both pipelines' AO rays on the third side stream, each behind its pre-pass; the marches start at once
          HIP_TRY(this, hipEventRecord(evPre2, side2));
          HIP_TRY(this, hipEventRecord(evPre, sideN));
          HIP_TRY(this, hipStreamWaitEvent(side4, evPre2, 0));
          HIP_TRY(this, EXA_FORM(launchAoRaysKd)(ah, nPreHeavy, side4));
          HIP_TRY(this, hipEventRecord(evAo2, side4));
          HIP_TRY(this, hipStreamWaitEvent(side4, evPre, 0));
          HIP_TRY(this, EXA_FORM(launchAoRaysKd)(ac, nPreCheap, side4));
          HIP_TRY(this, hipEventRecord(evAo, side4));
        } else {
          HIP_TRY(this, EXA_FORM(launchAoRaysKd)(ah, nPreHeavy, side2));
          HIP_TRY(this, EXA_FORM(launchAoRaysKd)(ac, nPreCheap, sideN));
        }
        HIP_TRY(this, march(ac, nPreCheap, true, 0, sideN));
        if (overlap) {
          HIP_TRY(this, hipStreamWaitEvent(sideN, evAo, 0));
          HIP_TRY(this, EXA_FORM(launchCompositeKd)(ac, nPreCheap, sideN));
        }
        HIP_TRY(this, hipEventRecord(evJoinN, sideN));
        HIP_TRY(this, march(ah, nPreHeavy, true, 0, side2));
        if (overlap) {
          HIP_TRY(this, hipStreamWaitEvent(side2, evAo2, 0));
          HIP_TRY(this, EXA_FORM(launchCompositeKd)(ah, nPreHeavy, side2));
        }
        HIP_TRY(this, hipEventRecord(evJoin2, side2));
        HIP_TRY(this, hipStreamWaitEvent(s, evJoin2, 0));
        HIP_TRY(this, hipStreamWaitEvent(s, evJoinN, 0));
      } else {
      if (surfOn) HIP_TRY(this, EXA_FORM(launchSurfacePrepassKd)(a, numBlocks, stats, s));
      if (surfOn && !stats && !overlap) HIP_TRY(this, EXA_FORM(launchAoRaysKd)(a, numBlocks, s));
      if (overlap) {
        HIP_TRY(this, hipEventRecord(evPre, s));
        HIP_TRY(this, hipStreamWaitEvent(side4, evPre, 0));
        HIP_TRY(this, EXA_FORM(launchAoRaysKd)(a, numBlocks, side4));
        HIP_TRY(this, hipEventRecord(evAo, side4));
        HIP_TRY(this, march(a, numBlocks, true, 0, s));
        HIP_TRY(this, hipStreamWaitEvent(s, evAo, 0));
        HIP_TRY(this, EXA_FORM(launchCompositeKd)(a, numBlocks, s));
      } else if (!wide) {
        HIP_TRY(this, march(a, numBlocks, surfOn, stats ? statsMode : 0, s));
      } else {
        // the critical tiles on side streams so that they start together with the rest of the frame
        HIP_TRY(this, hipEventRecord(evFork, s));
        RenderArgs aw = a;
        if (nWide4) {
          HIP_TRY(this, hipStreamWaitEvent(side4, evFork, 0));
          aw.wideTileMap = wideMap.p;
          aw.wideSegs = wideSegs.p;
          HIP_TRY(this, EXA_FORM(launchRenderKdWide)(aw, nWide4, lanesTopInUse, p.gradientShadingDVR != 0, fastMath != 0, surfOn, side4));
          HIP_TRY(this, hipEventRecord(evJoin4, side4));
        }
        if (nWide2) {
          HIP_TRY(this, hipStreamWaitEvent(side2, evFork, 0));
          aw.wideTileMap = wideMap.p + nWide4;
          aw.wideSegs = wideSegs.p + size_t(nWide4) * segsPerRay(lanesTopInUse) * kTilePixels;
          HIP_TRY(this, EXA_FORM(launchRenderKdWide)(aw, nWide2, 2, p.gradientShadingDVR != 0, fastMath != 0, surfOn, side2));
          HIP_TRY(this, hipEventRecord(evJoin2, side2));
        }
        // the rest of the frame on a stream of its own as well: launched on the caller's stream it would not
        // overlap the side streams when that stream is the (synchronising) null stream
        RenderArgs an = a;
        an.tileMap = normalMap.p;
        HIP_TRY(this, hipStreamWaitEvent(sideN, evFork, 0));
        HIP_TRY(this, march(an, nNormal, surfOn, 0, sideN));
        HIP_TRY(this, hipEventRecord(evJoinN, sideN));
        if (nWide4) HIP_TRY(this, hipStreamWaitEvent(s, evJoin4, 0));
        if (nWide2) HIP_TRY(this, hipStreamWaitEvent(s, evJoin2, 0));
        HIP_TRY(this, hipStreamWaitEvent(s, evJoinN, 0));
      }
      }
    }
    else         HIP_TRY(this, EXA_FORM(launchRender)(a, numBlocks, p.gradientShadingDVR != 0, surfacesEnabled(), stats, s));
    last.node_bytes = useKd() ? sizeof(KdNodeDev) : sizeof(BvhNode);
    HIP_TRY(this, hipEventRecord(ev1, s));
    return 0;
  }
};

// Leaves and neighbour links of the rope walk (exa_ropes.h), from the region kd-tree and the regions' domains as the device
// holds them; this adds what the march needs of a region (its packed record) and uploads the result.
int ExaHipRenderer::buildRopes()
{
  const auto tBuild0 = std::chrono::steady_clock::now();
  const size_t nk = kdNodes.n, nr = sc.numRegions;
  static_assert(sizeof(KdNodeDev) == sizeof(ExaKdNode), "the device's kd node is the ABI's with activity bits in the axis word");
  std::vector<ExaKdNode> kd(nk);
  std::vector<float> dom(6 * nr);
  std::vector<RegionInfo> ri(nr);
  if (nk) HIP_TRY(this, hipMemcpy(kd.data(), kdNodes.p, nk * sizeof(KdNodeDev), hipMemcpyDeviceToHost));
  HIP_TRY(this, hipMemcpy(dom.data(), domain.p, dom.size() * sizeof(float), hipMemcpyDeviceToHost));
  HIP_TRY(this, hipMemcpy(ri.data(), regionInfo.p, nr * sizeof(RegionInfo), hipMemcpyDeviceToHost));
  const unsigned nthreads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  RopeBuild rb;
  buildRopesHost(kd.data(), nk, kdRoot, dom.data(), nr, kdLo, kdHi, nthreads, rb);
  if (rb.leaves.size() >= 0x7ffffff0ull) { ropeFailed = true; return 0; }
  if (!rb.boxesMatch) {
    // a tree whose planes do not reproduce the regions' domains (a caller's own kd-tree): the stack walk stays
    if (std::getenv("EXA_HIP_VERBOSE")) std::fprintf(stderr, "[exa_hip] rope walk: the kd-tree's planes do not reproduce the region domains; stack walk kept\n");
    ropeFailed = true;
    return 0;
  }
  std::vector<RopeLeaf> leaves(rb.leaves.size());
  const uint32_t bb = leafBeginBits, sb = leafSizeBits;
  for (size_t id = 0; id < leaves.size(); id++) {
    const RopeLeafHost &H = rb.leaves[id];
    RopeLeaf &L = leaves[id];
    L.lo[0] = H.lo[0]; L.lo[1] = H.lo[1]; L.lo[2] = H.lo[2];
    L.hi0 = H.hi[0]; L.hi1 = H.hi[1]; L.hi2 = H.hi[2];
    for (int f = 0; f < 6; f++) L.rope[f] = H.rope[f];
    L.flags = 0; L.pad = 0;
    L.region = H.region;
    L.rec = H.region >= 0 ? (uint32_t)H.region : 0u;
    if (H.region >= 0 && bb) {
      int lv = 0;
      while (float(1 << lv) < ri[id].finestLevelCellWidth) lv++;
      L.rec = uint32_t(ri[id].listBegin) | (uint32_t(ri[id].listSize - 1) << bb) | (uint32_t(lv) << (bb + sb));
    }
  }
  static_assert(sizeof(ExaKdNode) == 16, "kd node = one 16-byte load");
  if (ropeLeaves.upload(leaves.data(), leaves.size()) != hipSuccess
      || ropeNodes.upload(reinterpret_cast<const KdNodeDev *>(rb.nodes.data()), rb.nodes.size()) != hipSuccess) {
    // the links are an optimisation (64 B per leaf + 16 B per node of extra memory): without them the stack walk
    (void)hipGetLastError();
    ropeLeaves.release(); ropeNodes.release();
    ropeFailed = true;
    return 0;
  }
  ropeRoot = kdRoot;
  ropeFastDiv = rb.planesOnGrid ? 1 : 0;
  ropeAddr32 = (leaves.size() * sizeof(RopeLeaf) < (1ull << 32) && rb.nodes.size() * sizeof(KdNodeDev) < (1ull << 32)) ? 1 : 0;
  ropeBuilt = true;
  ropeFlagsStale = true;
  if (std::getenv("EXA_HIP_VERBOSE"))
    std::fprintf(stderr, "[exa_hip] rope walk: %zu leaves (%zu gaps), %zu nodes linked on %u threads in %.1f ms; short division %s\n", leaves.size(), rb.gaps,
                 rb.nodes.size(), nthreads, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tBuild0).count(), rb.planesOnGrid ? "on" : "off");
  return 0;
}

extern "C" {

const char *exa_hip_last_error(const ExaHipRenderer *h) { return h ? h->err.c_str() : g_createError.c_str(); }

int exa_hip_create(const ExaHipScene *scene, int32_t device, ExaHipRenderer **out)
{
  if (!out || !scene) { g_createError = "exa_hip_create: null argument"; return 1; }
  *out = nullptr;
  if (scene->allowEmptyCells != 0 && scene->allowEmptyCells != 1) {
    g_createError = "exa_hip_create: ExaHipScene.allowEmptyCells is 0 or 1 (was the struct zero-initialised before it was filled?)";
    return 1;
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_createError = std::string("exa_hip_create: no HIP device available (") + hipGetErrorString(e)
                  + "); this module has no CPU fallback";
    return 2;
  }
  if (device < 0 || device >= ndev) { g_createError = "exa_hip_create: bad device index"; return 1; }
  if (scene->numFields < 1 || scene->numFields > EXA_MAX_CHANNELS) { g_createError = "exa_hip_create: 1..10 scalar fields required"; return 1; }
  if (scene->numRegions == 0 || scene->numBricks == 0) { g_createError = "exa_hip_create: empty scene"; return 1; }
  if (scene->numRegions > 0x7fffffffull) { g_createError = "exa_hip_create: too many regions"; return 1; }
  ExaHipRenderer *h = new ExaHipRenderer;
  h->device = device;
  auto bail = [&]() { g_createError = h->err; delete h; return 1; };
#define CREATE_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { h->fail(std::string(#call) + ": " + hipGetErrorString(e_)); return bail(); } } while (0)
  DeviceGuard guard_(device);
  CREATE_TRY(guard_.err);

  // validate indices on the host before anything can fault on the device
  for (uint64_t i = 0; i < scene->leafListSize; i++)
    if (scene->leafList[i] < 0 || uint64_t(scene->leafList[i]) >= scene->numBricks) { h->fail("exa_hip_create: leaf list entry out of range"); return bail(); }
  for (uint64_t b = 0; b < scene->numBricks; b++) {
    const ExaBrick &B = scene->bricks[b];
    const uint64_t vol = uint64_t(B.size[0]) * uint64_t(B.size[1]) * uint64_t(B.size[2]);
    if (B.size[0] <= 0 || B.size[1] <= 0 || B.size[2] <= 0 || B.level < 0 || B.level > 30
        || uint64_t(B.begin) + vol > scene->totalCells) { h->fail("exa_hip_create: brick record out of range"); return bail(); }
  }
  for (int f = 0; f < scene->numFields; f++)
    if (scene->channelOffset[f] + scene->totalCells > uint64_t(scene->numFields) * scene->totalCells) { h->fail("exa_hip_create: channel offset out of range"); return bail(); }

  h->numFields = scene->numFields;
  h->emptyCells = scene->allowEmptyCells != 0;
  if (h->emptyCells) h->basisForm = 0;
  h->totalCells = scene->totalCells;
  h->numBricks = scene->numBricks; h->leafListSize = scene->leafListSize;
  {
    // brick orders: as uploaded, and along a Morton curve of the brick centres (21 bits per axis over the voxel bounds;
    // equal codes keep the uploaded order), with a running `begin` as the reference assigns it (OptixRenderer.cpp:71-93)
    const uint64_t nb = scene->numBricks;
    h->beginUploaded.resize(nb);
    std::vector<std::pair<uint64_t, uint32_t>> keyed(nb);
    for (uint64_t b = 0; b < nb; b++) {
      const ExaBrick &B = scene->bricks[b];
      h->beginUploaded[b] = B.begin;
      uint64_t code = 0;
      for (int k = 0; k < 3; k++) {
        const double cw = double(1u << B.level);
        const double c = double(B.lower[k]) + 0.5 * cw * double(B.size[k]);
        const double ext = double(scene->voxelBounds_hi[k]) - double(scene->voxelBounds_lo[k]);
        double u = ext > 0 ? (c - double(scene->voxelBounds_lo[k])) / ext : 0.0;
        u = std::min(std::max(u, 0.0), 1.0);
        code |= LbvhTopology::spread21(std::min<uint64_t>(uint64_t(u * 2097152.0), 2097151ull)) << k;
      }
      keyed[b] = { code, uint32_t(b) };
    }
    std::sort(keyed.begin(), keyed.end());
    h->beginMorton.resize(nb);
    uint64_t at = 0;
    for (uint64_t i = 0; i < nb; i++) {
      const ExaBrick &B = scene->bricks[keyed[i].second];
      h->beginMorton[keyed[i].second] = uint32_t(at);
      at += uint64_t(B.size[0]) * uint64_t(B.size[1]) * uint64_t(B.size[2]);
    }
    // The permutation moves field f at f * totalCells and brick b's cells as one block [begin, begin + volume): it needs the
    // layout the reference's constructor produces (OptixRenderer.cpp:71-110) — channel offsets f * totalCells and the
    // uploaded begins a partition of [0, totalCells) into the bricks' volumes.  Anything else keeps the uploaded order.
    bool partition = at == scene->totalCells;
    for (int f = 0; f < scene->numFields && partition; f++) partition = scene->channelOffset[f] == uint64_t(f) * scene->totalCells;
    if (partition) {
      std::vector<std::pair<uint32_t, uint64_t>> spans(nb);                  // (begin, volume), sorted by begin
      for (uint64_t b = 0; b < nb; b++) {
        const ExaBrick &B = scene->bricks[b];
        spans[b] = { B.begin, uint64_t(B.size[0]) * uint64_t(B.size[1]) * uint64_t(B.size[2]) };
      }
      std::sort(spans.begin(), spans.end());
      uint64_t run = 0;
      for (uint64_t b = 0; b < nb && partition; b++) { partition = spans[b].first == run; run += spans[b].second; }
    }
    h->brickOrderPossible = partition;
    if (!partition) h->beginMorton = h->beginUploaded;
    if (const char *e = std::getenv("EXA_BRICK_ORDER")) h->brickOrderWanted = std::atoi(e) != 0 && h->brickOrderPossible;
    if (const char *e = std::getenv("EXA_BASIS_FORM")) h->basisForm = std::atoi(e) != 0 && !h->emptyCells;    // initial value of option basis_form
  }
  for (int k = 0; k < 3; k++) { h->voxLo[k] = scene->voxelBounds_lo[k]; h->voxHi[k] = scene->voxelBounds_hi[k]; }
  static_assert(sizeof(ExaBrick) == 2 * sizeof(int4), "brick = two int4");
  CREATE_TRY(h->bricks.upload(reinterpret_cast<const int4 *>(scene->bricks), scene->numBricks * 2));
  CREATE_TRY(h->leafList.upload(scene->leafList, scene->leafListSize));
  {
    // march headers along the leaf list (the kd march reads the record at listBegin + child, no id indirection).
    // What a brick visit needs, ready to use — float(lower) (the conversion the reference's
    // `vec3f(brick.lower)` performs, exabrick.cu:623), 2^-level, the sizes and the first cell's offset
    std::vector<ExaBrick> hdr(scene->leafListSize);
    for (uint64_t i = 0; i < scene->leafListSize; i++) {
      const ExaBrick &B = scene->bricks[scene->leafList[i]];
      const float lowerF[3] = { float(B.lower[0]), float(B.lower[1]), float(B.lower[2]) };
      const float invCw = std::ldexp(1.f, -B.level);
      ExaBrick &o = hdr[i];
      std::memcpy(&o.lower[0], lowerF, sizeof(lowerF));
      std::memcpy(&o.size[0], &invCw, sizeof(float));
      o.size[1] = B.size[0]; o.size[2] = B.size[1]; o.level = B.size[2]; o.begin = B.begin;
    }
    CREATE_TRY(h->leafHdr.upload(reinterpret_cast<const int4 *>(hdr.data()), hdr.size() * 2));
  }
  // 24-bit multiplies in the cell address need every factor below 2^24 and every product below 2^32; 32-bit byte
  // offsets need a field below 4 GiB (the pair load reads one float past a row's last cell at most)
  h->mul24 = 1;
  for (uint64_t b = 0; b < scene->numBricks; b++) {
    const ExaBrick &B = scene->bricks[b];
    if (uint64_t(B.size[0]) * uint64_t(B.size[1]) >= (1ull << 24) || B.size[0] >= (1 << 24) || B.size[1] >= (1 << 24) || B.size[2] >= (1 << 24))
      h->mul24 = 0;
  }
  if (scene->totalCells >= (1ull << 32)) h->mul24 = 0;
  h->addr32 = ((scene->totalCells + 2) * sizeof(float) <= (1ull << 32)              // cell scalars of one field
               && scene->leafListSize * 32ull < (1ull << 32)                          // march headers
               && scene->numKdNodes * sizeof(KdNodeDev) < (1ull << 32)) ? 1 : 0;      // kd nodes
  CREATE_TRY(h->scalars.upload(scene->scalars, size_t(scene->numFields) * scene->totalCells));
  std::vector<RegionInfo> ri(scene->numRegions);
  std::vector<float2> vr(scene->numRegions);
  std::vector<float> dom(scene->numRegions * 6);
  for (uint64_t r = 0; r < scene->numRegions; r++) {
    const ExaBrickRegion &R = scene->regions[r];
    if (R.leafListSize < 1 || R.leafListBegin < 0 || uint64_t(R.leafListBegin) + uint64_t(R.leafListSize) > scene->leafListSize) {
      h->fail("exa_hip_create: region leaf list out of range"); return bail();
    }
    // finestLevelCellWidth = 2^(min level) (exa/Regions.cpp:293-299): the kernels rely on an integer-valued width >= 1
    // and on a power of two (the reference only ever writes 1 << finestLevel): the march forms 1/(dt*width) from the
    // width's exponent bits
    {
      int ex = 0;
      const float mant = std::frexp(R.finestLevelCellWidth, &ex);
      if (!(R.finestLevelCellWidth >= 1.f && R.finestLevelCellWidth <= 1073741824.f) || mant != 0.5f) {
        h->fail("exa_hip_create: region finestLevelCellWidth is not a power of two >= 1"); return bail();
      }
    }
    ri[r].listBegin = R.leafListBegin;
    ri[r].listSize = R.leafListSize;
    ri[r].finestLevelCellWidth = R.finestLevelCellWidth;
    ri[r].firstBrick = scene->leafList[R.leafListBegin];
    vr[r] = make_float2(R.valueRange_lo, R.valueRange_hi);
    for (int k = 0; k < 3; k++) { dom[6 * r + k] = R.domain_lo[k]; dom[6 * r + 3 + k] = R.domain_hi[k]; }
  }
  CREATE_TRY(h->regionInfo.upload(ri.data(), ri.size()));
  CREATE_TRY(h->valueRange.upload(vr.data(), vr.size()));
  CREATE_TRY(h->domain.upload(dom.data(), dom.size()));

  // ---- optional region kd-tree: validate, order by height for the refit, upload ----
  for (int k = 0; k < 3; k++) { h->kdLo[k] = INFINITY; h->kdHi[k] = -INFINITY; }
  for (uint64_t r = 0; r < scene->numRegions; r++)
    for (int k = 0; k < 3; k++) {
      h->kdLo[k] = std::fmin(h->kdLo[k], scene->regions[r].domain_lo[k]);
      h->kdHi[k] = std::fmax(h->kdHi[k], scene->regions[r].domain_hi[k]);
    }
  if (scene->kdNodes != nullptr || (scene->numKdNodes == 0 && scene->numRegions == 1 && scene->kdRoot == ~int32_t(0))) {
    const uint64_t nk = scene->numKdNodes;
    auto refOk = [&](int32_t ref) {
      if (ref == EXA_KD_EMPTY) return true;
      return ref >= 0 ? uint64_t(ref) < nk : uint64_t(~ref) < scene->numRegions;
    };
    bool ok = refOk(scene->kdRoot) && scene->kdRoot != EXA_KD_EMPTY && nk < 0x7fffffffull;
    for (uint64_t i = 0; ok && i < nk; i++) {
      const ExaKdNode &n = scene->kdNodes[i];
      // children must come later in the array (preorder), which also rules out cycles
      ok = n.axis >= 0 && n.axis <= 2 && refOk(n.left) && refOk(n.right)
           && (n.left < 0 || uint64_t(n.left) > i) && (n.right < 0 || uint64_t(n.right) > i);
    }
    if (!ok) { h->fail("exa_hip_create: malformed kd-tree"); return bail(); }
    std::vector<int32_t> kh(nk, 1);
    for (uint64_t ii = nk; ii-- > 0;) {            // children have larger indices: one backward sweep
      const ExaKdNode &n = scene->kdNodes[ii];
      int hh = 0;
      if (n.left >= 0) hh = std::max(hh, kh[n.left]);
      if (n.right >= 0) hh = std::max(hh, kh[n.right]);
      kh[ii] = hh + 1;
    }
    int kmax = 0;
    for (uint64_t i = 0; i < nk; i++) kmax = std::max(kmax, kh[i]);
    std::vector<int> kcount(kmax + 2, 0);
    for (uint64_t i = 0; i < nk; i++) kcount[kh[i]]++;
    h->kdLevelBegin.assign(1, 0);
    for (int hh = 1; hh <= kmax; hh++) h->kdLevelBegin.push_back(h->kdLevelBegin.back() + kcount[hh]);
    std::vector<int32_t> kids(nk);
    {
      std::vector<int> cursor(h->kdLevelBegin.begin(), h->kdLevelBegin.end());
      for (uint64_t i = 0; i < nk; i++) kids[cursor[kh[i] - 1]++] = (int32_t)i;
    }
    std::vector<KdNodeDev> kd(nk);
    for (uint64_t i = 0; i < nk; i++) {
      kd[i].split = scene->kdNodes[i].split;
      kd[i].word = (uint32_t)scene->kdNodes[i].axis;
      kd[i].left = scene->kdNodes[i].left;
      kd[i].right = scene->kdNodes[i].right;
    }
    std::vector<RegionRec> rec(scene->numRegions);
    for (uint64_t r = 0; r < scene->numRegions; r++) {
      const ExaBrickRegion &R = scene->regions[r];
      RegionRec &q = rec[r];
      q.lo[0] = R.domain_lo[0]; q.lo[1] = R.domain_lo[1]; q.lo[2] = R.domain_lo[2];
      q.hi0 = R.domain_hi[0]; q.hi1 = R.domain_hi[1]; q.hi2 = R.domain_hi[2];
      q.finestLevelCellWidth = R.finestLevelCellWidth;
      q.firstBrick = scene->leafList[R.leafListBegin];
      q.listBegin = R.leafListBegin; q.listSize = R.leafListSize; q.pad0 = q.pad1 = 0;
    }
    CREATE_TRY(h->kdNodes.upload(kd.data(), kd.size()));
    // March tree: the same nodes with every leaf reference replaced by the region's record
    // {listBegin | listSize-1 | log2(finestLevelCellWidth)}, when the scene's ranges fit 31 bits, so that a segment
    // start needs no region-info load (one dependent HBM/L2 round trip less per segment).
    {
      auto bitsFor = [](uint64_t maxValue) { uint32_t b = 0; while (b < 63 && (1ull << b) <= maxValue) b++; return b; };
      uint64_t maxSize = 1; int maxLevel = 0; bool pow2 = true;
      for (uint64_t r = 0; r < scene->numRegions; r++) {
        const ExaBrickRegion &R = scene->regions[r];
        maxSize = std::max<uint64_t>(maxSize, (uint64_t)R.leafListSize);
        int lv = 0;
        while (lv < 31 && float(1 << lv) < R.finestLevelCellWidth) lv++;
        if (float(1 << lv) != R.finestLevelCellWidth) pow2 = false;
        maxLevel = std::max(maxLevel, lv);
      }
      const uint32_t bb = std::max(1u, bitsFor(scene->leafListSize ? scene->leafListSize - 1 : 0));
      const uint32_t sb = bitsFor(maxSize - 1), lb = bitsFor((uint64_t)maxLevel);
      if (pow2 && bb + sb + lb <= 31) {
        auto pack = [&](int32_t ref) -> int32_t {
          if (ref >= 0 || ref == EXA_KD_EMPTY) return ref;
          const ExaBrickRegion &R = scene->regions[~ref];
          int lv = 0;
          while (float(1 << lv) < R.finestLevelCellWidth) lv++;
          const uint32_t d = uint32_t(R.leafListBegin) | (uint32_t(R.leafListSize - 1) << bb) | (uint32_t(lv) << (bb + sb));
          return ~int32_t(d);
        };
        std::vector<KdNodeDev> mk(kd);
        bool clash = false;
        for (auto &n : mk) {
          n.left = pack(n.left); n.right = pack(n.right);
          // ~d must not collide with the walk's two sentinels (INT32_MIN, INT32_MIN + 1)
          clash = clash || (n.left < 0 && n.left != EXA_KD_EMPTY && n.left <= INT32_MIN + 1) || (n.right < 0 && n.right != EXA_KD_EMPTY && n.right <= INT32_MIN + 1);
        }
        const int32_t root = pack(scene->kdRoot);
        clash = clash || (root < 0 && root <= INT32_MIN + 1);
        if (!clash) {
          if (mk.empty()) mk.resize(1);              // single-region scene: the root is the leaf
          CREATE_TRY(h->kdMarchNodes.upload(mk.data(), mk.size()));
          h->kdMarchRoot = root;
          h->leafBeginBits = bb; h->leafSizeBits = sb;
        }
      }
    }
    CREATE_TRY(h->kdLevelIds.upload(kids.data(), kids.size()));
    CREATE_TRY(h->regionRec.upload(rec.data(), rec.size()));
    h->kdRoot = scene->kdRoot;
    h->haveKd = true;
  }
  CREATE_TRY(h->volActive.alloc(scene->numRegions));
  CREATE_TRY(h->isoActive.alloc(scene->numRegions));
  CREATE_TRY(h->xf.alloc(size_t(EXA_MAX_CHANNELS) * EXA_NUM_XF_VALUES));
  std::memset(h->xfHost, 0, sizeof(h->xfHost));
  CREATE_TRY(h->statsBuf.alloc(ST_COUNT));
  CREATE_TRY(h->errorFlag.alloc(1));
  CREATE_TRY(hipMemset(h->errorFlag.p, 0, sizeof(int32_t)));
  CREATE_TRY(hipEventCreate(&h->ev0));
  CREATE_TRY(hipEventCreate(&h->ev1));
  CREATE_TRY(hipEventCreate(&h->ev2));
  CREATE_TRY(hipEventCreateWithFlags(&h->evFork, hipEventDisableTiming));
  CREATE_TRY(hipEventCreateWithFlags(&h->evJoin4, hipEventDisableTiming));
  CREATE_TRY(hipEventCreateWithFlags(&h->evJoin2, hipEventDisableTiming));
  CREATE_TRY(hipStreamCreateWithFlags(&h->side4, hipStreamNonBlocking));
  CREATE_TRY(hipStreamCreateWithFlags(&h->side2, hipStreamNonBlocking));
  CREATE_TRY(hipStreamCreateWithFlags(&h->sideN, hipStreamNonBlocking));
  CREATE_TRY(hipEventCreateWithFlags(&h->evJoinN, hipEventDisableTiming));
  CREATE_TRY(hipEventCreateWithFlags(&h->evPre, hipEventDisableTiming));
  CREATE_TRY(hipEventCreateWithFlags(&h->evPre2, hipEventDisableTiming));
  CREATE_TRY(hipEventCreateWithFlags(&h->evAo, hipEventDisableTiming));
  CREATE_TRY(hipEventCreateWithFlags(&h->evAo2, hipEventDisableTiming));
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
      h->numSimdWaves = prop.multiProcessorCount * 4 * 6;
  }

  h->sc.bricks = h->bricks.p;
  h->sc.leafList = h->leafList.p;
  h->sc.leafHdr = h->leafHdr.p;
  h->sc.scalars = h->scalars.p;
  h->sc.regionInfo = h->regionInfo.p;
  h->sc.valueRange = h->valueRange.p;
  h->sc.domain = h->domain.p;
  for (int f = 0; f < EXA_MAX_CHANNELS; f++) h->sc.channelOffset[f] = f < scene->numFields ? scene->channelOffset[f] : 0;
  h->sc.numRegions = (uint32_t)scene->numRegions;
  h->sc.numInternal = 0;                     // set when the LBVH is built (ensureLbvh)
#undef CREATE_TRY
  *out = h;
  return 0;
}

// One handle, several devices: the scene is replicated, device i renders the 16x16 tiles t with t % n == i and stores
// them straight into the destination frame on the first device of the list (peer-mapped when it is another device), so
// there is no gather and no untile step.  Entries of `devices` may repeat (several renderers sharing one GPU: rehearsal).
int exa_hip_create_multi(const ExaHipScene *scene, const int32_t *devices, int32_t numDevices, ExaHipRenderer **out)
{
  if (!out || !scene || !devices || numDevices < 1 || numDevices > 64) { g_createError = "exa_hip_create_multi: bad arguments"; return 1; }
  *out = nullptr;
  ExaHipRenderer *h = new ExaHipRenderer;
  h->device = devices[0];
  auto bail = [&](const std::string &msg) { g_createError = msg; exa_hip_destroy(h); return 1; };
  for (int i = 0; i < numDevices; i++) {
    ExaHipRenderer *c = nullptr;
    if (int rc = exa_hip_create(scene, devices[i], &c)) { exa_hip_destroy(h); return rc; }    // g_createError is set
    h->children.push_back(c);
    c->colorRowMajor = true;
    c->rank = i; c->world = numDevices; c->layoutDirty = true;
    DeviceGuard g(devices[i]);
    if (g.err != hipSuccess || hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking) != hipSuccess)
      return bail("exa_hip_create_multi: cannot create a stream on device " + std::to_string(devices[i]));
    if (devices[i] != devices[0]) {
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, devices[i], devices[0]) != hipSuccess || !can)
        return bail("exa_hip_create_multi: device " + std::to_string(devices[i]) + " cannot access device " + std::to_string(devices[0]));
      const hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return bail(std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
      (void)hipGetLastError();
    }
  }
  {
    DeviceGuard g(devices[0]);
    if (g.err != hipSuccess || hipEventCreateWithFlags(&h->evCall, hipEventDisableTiming) != hipSuccess) return bail("exa_hip_create_multi: hipEventCreate failed");
  }
  h->numFields = h->children[0]->numFields;
  *out = h;
  return 0;
}

int exa_hip_destroy(ExaHipRenderer *h)
{
  if (!h) return 0;
  for (ExaHipRenderer *c : h->children) exa_hip_destroy(c);
  h->children.clear();
  DeviceGuard guard_(h->device);
  if (h->ownStream) (void)hipStreamDestroy(h->ownStream);
  if (h->evCall) (void)hipEventDestroy(h->evCall);
  (void)hipDeviceSynchronize();
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev2) (void)hipEventDestroy(h->ev2);
  if (h->evFork) (void)hipEventDestroy(h->evFork);
  if (h->evJoin4) (void)hipEventDestroy(h->evJoin4);
  if (h->evJoin2) (void)hipEventDestroy(h->evJoin2);
  if (h->side4) (void)hipStreamDestroy(h->side4);
  if (h->side2) (void)hipStreamDestroy(h->side2);
  if (h->sideN) (void)hipStreamDestroy(h->sideN);
  if (h->evJoinN) (void)hipEventDestroy(h->evJoinN);
  for (hipEvent_t e : { h->evPre, h->evPre2, h->evAo, h->evAo2 }) if (e) (void)hipEventDestroy(e);
  delete h;
  return 0;
}

int exa_hip_resize(ExaHipRenderer *h, int32_t width, int32_t height)
{
  if (h && !h->children.empty()) {
    if (width <= 0 || height <= 0 || int64_t(width) * height > (int64_t(1) << 30)) { h->fail("exa_hip_resize: bad size"); return 1; }
    for (ExaHipRenderer *c : h->children)
      if (int rc = exa_hip_resize(c, width, height)) { h->fail(c->err); return rc; }
    h->W = width; h->H = height;
    EXA_ON_DEVICE(h);                         // the root device holds the frame a host destination is copied from
    HIP_TRY(h, h->color.alloc(size_t(width) * height));
    return 0;
  }
  if (!h) return 1;
  if (width <= 0 || height <= 0 || int64_t(width) * height > (int64_t(1) << 30)) { h->fail("exa_hip_resize: bad size"); return 1; }
  EXA_ON_DEVICE(h);
  h->W = width; h->H = height;
  h->layoutDirty = true;
  return h->rebuildLayout();
}

int exa_hip_set_frame_state(ExaHipRenderer *h, const ExaHipFrameState *fs)
{
  if (h && !h->children.empty()) {            // multi-device handle: the same call on every device's renderer
    for (ExaHipRenderer *c : h->children)
      if (int rc = exa_hip_set_frame_state(c, fs)) { h->fail(c->err); return rc; }
    return 0;
  }
  if (!h || !fs) return 1;
  if (!h->haveFs || std::memcmp(h->fs.xfDomain, fs->xfDomain, sizeof(fs->xfDomain)) != 0
      || h->fs.xfOpacityScale != fs->xfOpacityScale) h->volDirty = true;
  if (!h->haveFs || std::memcmp(h->fs.iso, fs->iso, sizeof(fs->iso)) != 0) h->isoDirty = true;
  if (h->haveFs) {
    // anything but the frame id changes which tiles are expensive
    ExaHipFrameState a = h->fs, b = *fs;
    a.frameID = b.frameID = 0;
    if (std::memcmp(&a, &b, sizeof(a)) != 0) h->costPhase = 1;
  }
  h->fs = *fs;
  h->haveFs = true;
  return 0;
}

int exa_hip_set_xf(ExaHipRenderer *h, int32_t chan, const float *rgba128)
{
  if (h && !h->children.empty()) {            // multi-device handle: the same call on every device's renderer
    for (ExaHipRenderer *c : h->children)
      if (int rc = exa_hip_set_xf(c, chan, rgba128)) { h->fail(c->err); return rc; }
    return 0;
  }
  if (!h || !rgba128) return 1;
  if (chan < 0 || chan >= EXA_MAX_CHANNELS) { h->fail("exa_hip_set_xf: bad channel"); return 1; }
  std::memcpy(h->xfHost[chan], rgba128, sizeof(h->xfHost[chan]));
  h->xfDirty = true;
  h->volDirty = true;                      // needVolumeBVHRebuild = true (OptixRenderer.cpp:403)
  h->costPhase = 1;
  return 0;
}

int exa_hip_set_triangles(ExaHipRenderer *h, const float *vertices, uint64_t numVertices,
                          const int32_t *triangles, uint64_t numTris)
{
  if (h && !h->children.empty()) {            // multi-device handle: the same call on every device's renderer
    for (ExaHipRenderer *c : h->children)
      if (int rc = exa_hip_set_triangles(c, vertices, numVertices, triangles, numTris)) { h->fail(c->err); return rc; }
    return 0;
  }
  if (!h) return 1;
  EXA_ON_DEVICE(h);
  HIP_TRY(h, hipDeviceSynchronize());
  h->numTris = 0;
  h->meshNodes.release(); h->meshVerts.release(); h->meshTris.release();
  if (numTris == 0) return 0;
  if (!vertices || !triangles || numTris > 0x3fffffffull) { h->fail("exa_hip_set_triangles: bad arguments"); return 1; }
  for (uint64_t i = 0; i < 3 * numTris; i++)
    if (triangles[i] < 0 || uint64_t(triangles[i]) >= numVertices) { h->fail("broken triangle model"); return 1; }   // TriangleMesh.cpp:43-52
  // boxes, slightly padded so that axis-aligned triangles keep a non-degenerate slab interval
  std::vector<float> boxes(6 * numTris);
  for (uint64_t t = 0; t < numTris; t++)
    for (int k = 0; k < 3; k++) {
      const float a = vertices[3 * triangles[3 * t] + k], b = vertices[3 * triangles[3 * t + 1] + k], c = vertices[3 * triangles[3 * t + 2] + k];
      const float lo = std::fmin(a, std::fmin(b, c)), hi = std::fmax(a, std::fmax(b, c));
      const float pad = 1e-5f * std::fmax(std::fmax(std::fabs(lo), std::fabs(hi)), hi - lo) + 1e-30f;
      boxes[6 * t + k] = lo - pad; boxes[6 * t + 3 + k] = hi + pad;
    }
  LbvhTopology topo;
  topo.build(boxes.data(), numTris);
  const size_t ni = topo.child0.size();
  std::vector<BvhNode> nodes(ni);
  std::vector<float> nlo(3 * ni), nhi(3 * ni);
  auto childBox = [&](int32_t c, float *lo, float *hi) {
    if (c == INT32_MIN) { for (int k = 0; k < 3; k++) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; } return; }
    if (c < 0) { for (int k = 0; k < 3; k++) { lo[k] = boxes[6 * size_t(~c) + k]; hi[k] = boxes[6 * size_t(~c) + 3 + k]; } return; }
    for (int k = 0; k < 3; k++) { lo[k] = nlo[3 * size_t(c) + k]; hi[k] = nhi[3 * size_t(c) + k]; }
  };
  for (size_t i = ni; i-- > 0;) {                       // children have larger indices than their parent
    float l0[3], h0[3], l1[3], h1[3];
    childBox(topo.child0[i], l0, h0);
    childBox(topo.child1[i], l1, h1);
    for (int k = 0; k < 3; k++) { nlo[3 * i + k] = std::fmin(l0[k], l1[k]); nhi[3 * i + k] = std::fmax(h0[k], h1[k]); }
    BvhNode &n = nodes[i];
    n.q0 = make_float4(l0[0], l0[1], l0[2], h0[0]);
    n.q1 = make_float4(h0[1], h0[2], l1[0], l1[1]);
    n.q2 = make_float4(l1[2], h1[0], h1[1], h1[2]);
    n.child0 = topo.child0[i]; n.child1 = topo.child1[i]; n.pad0 = n.pad1 = 0;
  }
  HIP_TRY(h, h->meshNodes.upload(nodes.data(), nodes.size()));
  HIP_TRY(h, h->meshVerts.upload(vertices, 3 * numVertices));
  HIP_TRY(h, h->meshTris.upload(triangles, 3 * numTris));
  h->numTris = (int)numTris;
  return 0;
}

int exa_hip_reset_tracer(ExaHipRenderer *h, const ExaHipTracer *t, const float *seeds)
{
  if (h && !h->children.empty()) {            // multi-device handle: the same call on every device's renderer
    for (ExaHipRenderer *c : h->children)
      if (int rc = exa_hip_reset_tracer(c, t, seeds)) { h->fail(c->err); return rc; }
    return 0;
  }
  if (!h || !t || !seeds) return 1;
  if (t->numTraces < 0 || t->numTimesteps < 2 || (long long)t->numTraces * t->numTimesteps > (1ll << 28)) { h->fail("exa_hip_reset_tracer: bad trace counts"); return 1; }
  for (int k = 0; k < 3; k++)
    if (t->channels[k] < 0 || t->channels[k] >= h->numFields) { h->fail("exa_hip_reset_tracer: tracer channel out of range"); return 1; }
  EXA_ON_DEVICE(h);
  HIP_TRY(h, hipDeviceSynchronize());
  h->tracer = *t;
  h->haveTracer = true;
  std::vector<float> host(size_t(t->numTraces) * t->numTimesteps * 3, 0.f);
  for (int i = 0; i < t->numTraces; i++) std::memcpy(&host[size_t(i) * t->numTimesteps * 3], &seeds[3 * i], 3 * sizeof(float));
  HIP_TRY(h, h->traces.upload(host.data(), host.size()));
  h->timestep = 0;
  h->streamDirty = true;                     // needStreamlineBVHRebuild = true (:461)
  return 0;
}

int exa_hip_set_tracer_enabled(ExaHipRenderer *h, int32_t enabled)
{
  if (h && !h->children.empty()) {            // multi-device handle: the same call on every device's renderer
    for (ExaHipRenderer *c : h->children)
      if (int rc = exa_hip_set_tracer_enabled(c, enabled)) { h->fail(c->err); return rc; }
    return 0;
  }
  if (!h) return 1;
  h->tracer.enabled = enabled;
  return 0;
}

int exa_hip_advance_tracer(ExaHipRenderer *h, int32_t *rebuild)
{
  if (h && !h->children.empty()) {
    for (size_t i = 0; i < h->children.size(); i++) {
      int32_t r = 0;
      if (int rc = exa_hip_advance_tracer(h->children[i], &r)) { h->fail(h->children[i]->err); return rc; }
      if (i == 0 && rebuild) *rebuild = r;
    }
    return 0;
  }
  if (!h) return 1;
  if (rebuild) *rebuild = 0;
  if (!h->haveTracer || !h->tracer.enabled) return 0;
  h->timestep++;
  if (h->timestep <= h->tracer.numTimesteps) h->streamDirty = true;
  if (rebuild) *rebuild = h->streamDirty;
  return 0;
}

int exa_hip_read_traces(ExaHipRenderer *h, float *dst)
{
  if (h && !h->children.empty()) {             // every device holds the same traces
    const int rc = exa_hip_read_traces(h->children[0], dst);
    if (rc) h->fail(h->children[0]->err);
    return rc;
  }
  if (!h || !dst || !h->haveTracer) return 1;
  EXA_ON_DEVICE(h);
  HIP_TRY(h, hipDeviceSynchronize());
  HIP_TRY(h, hipMemcpy(dst, h->traces.p, h->traces.n * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int exa_hip_set_params(ExaHipRenderer *h, const ExaHipParams *p)
{
  if (h && !h->children.empty()) {            // multi-device handle: the same call on every device's renderer
    for (ExaHipRenderer *c : h->children)
      if (int rc = exa_hip_set_params(c, p)) { h->fail(c->err); return rc; }
    return 0;
  }
  if (!h || !p) return 1;
  if (!(p->dt > 0.f)) { h->fail("exa_hip_set_params: dt must be > 0"); return 1; }
  if (!h->haveParams || h->p.numChannels != p->numChannels || h->p.spaceSkippingEnabled != p->spaceSkippingEnabled)
    h->volDirty = true;
  if (!h->haveParams || std::memcmp(&h->p, p, sizeof(*p)) != 0) h->costPhase = 1;
  h->p = *p;
  h->haveParams = true;
  return 0;
}

int exa_hip_set_shard(ExaHipRenderer *h, int32_t rank, int32_t worldSize)
{
  if (h && !h->children.empty()) { h->fail("exa_hip_set_shard: a multi-device handle shards the frame internally"); return 1; }
  if (!h) return 1;
  if (worldSize < 1 || rank < 0 || rank >= worldSize) { h->fail("exa_hip_set_shard: bad rank/world"); return 1; }
  h->rank = rank; h->world = worldSize;
  h->layoutDirty = true;
  if (h->W > 0) { EXA_ON_DEVICE(h); return h->rebuildLayout(); }
  return 0;
}

int exa_hip_set_option(ExaHipRenderer *h, const char *key, int32_t value)
{
  if (h && !h->children.empty()) {            // multi-device handle: the same call on every device's renderer
    for (ExaHipRenderer *c : h->children)
      if (int rc = exa_hip_set_option(c, key, value)) { h->fail(c->err); return rc; }
    return 0;
  }
  if (!h || !key) return 1;
  if (!std::strcmp(key, "tile_order")) { h->tileOrder = value; h->layoutDirty = true; return 0; }
  if (!std::strcmp(key, "tile_feedback")) { h->feedback = value; h->layoutDirty = true; return 0; }
  if (!std::strcmp(key, "wide_march")) {
    if (value != 0 && value != 1 && value != 2 && value != 4) { h->fail("exa_hip_set_option: wide_march is 0, 1, 2 or 4"); return 1; }
    h->wideMode = value; h->layoutDirty = true; return 0;
  }
  if (!std::strcmp(key, "stats_mode")) {
    if (value != 1 && value != 2) { h->fail("exa_hip_set_option: stats_mode is 1 or 2"); return 1; }
    h->statsMode = value; return 0;
  }
  if (!std::strcmp(key, "ao_defer")) {
    if (value < 0 || value > 2) { h->fail("exa_hip_set_option: ao_defer is 0, 1 or 2"); return 1; }
    h->aoDefer = value; return 0;
  }
  if (!std::strcmp(key, "prepass_split")) { h->prepassSplit = value != 0; h->costPhase = 1; return 0; }
  if (!std::strcmp(key, "ao_overlap")) { h->aoOverlap = value != 0; return 0; }
  if (!std::strcmp(key, "walk_probe")) { h->walkProbeOn = value != 0; return 0; }
  if (!std::strcmp(key, "debug_pixel")) { h->debugPixel = value; return 0; }
  if (!std::strcmp(key, "profile_marker")) {            // an empty kernel on the null stream, visible in a profiler's dispatch list
    EXA_ON_DEVICE(h);
    if (launchProfileMarker(value, nullptr) != hipSuccess) { h->fail("exa_hip_set_option: profile_marker launch failed"); return 1; }
    return 0;
  }
  if (!std::strcmp(key, "accel")) { h->accel = value; return 0; }
  if (!std::strcmp(key, "walk")) {
    if (value < 0 || value > 2) { h->fail("exa_hip_set_option: walk is 0 (chosen per frame), 1 (stack walk) or 2 (rope walk)"); return 1; }
    h->walkMode = value; return 0;
  }
  if (!std::strcmp(key, "lbvh_build")) {              // 0 = on the device (default), 1 = on the host; before the first LBVH frame
    if (h->lbvhBuilt && value != h->lbvhOnHost) { h->fail("exa_hip_set_option: lbvh_build must be set before the LBVH is first used"); return 1; }
    h->lbvhOnHost = value; return 0;
  }
  if (!std::strcmp(key, "fast_math")) { h->fastMath = value; return 0; }
  if (!std::strcmp(key, "fast_sampler")) { h->fastSampler = value; return 0; }
  if (!std::strcmp(key, "basis_form")) {
    if (value != 0 && value != 1) { h->fail("exa_hip_set_option: basis_form is 0 or 1"); return 1; }
    if (value && h->emptyCells) {
      h->fail("exa_hip_set_option: a scene with empty cells keeps basis_form 0 (an empty cell is a per-corner property, the per-axis association needs per-axis ones)");
      return 1;
    }
    h->basisForm = value; return 0;
  }
  if (!std::strcmp(key, "interleave")) { h->interleave = value != 0; return 0; }
  if (!std::strcmp(key, "addr64")) { h->addr64 = value != 0; return 0; }
  if (!std::strcmp(key, "pack_records")) { h->packRecords = value != 0; return 0; }
  if (!std::strcmp(key, "brick_order")) {
    if (value && !h->brickOrderPossible) {
      h->fail("exa_hip_set_option: brick_order 1 needs channel offsets f * totalCells and brick begins that partition [0, totalCells)");
      return 1;
    }
    h->brickOrderWanted = value != 0; return 0;
  }
  if (!std::strcmp(key, "tf_filter")) {
    if (value != 0 && value != 1) { h->fail("exa_hip_set_option: tf_filter is 0 or 1"); return 1; }
    if (value != h->tfFilter) { h->tfFilter = value; h->volDirty = true; }     // region activity goes through the TF lookup
    return 0;
  }
  h->fail(std::string("exa_hip_set_option: unknown key ") + key);
  return 1;
}

uint64_t exa_hip_output_pixels(const ExaHipRenderer *h)
{
  if (!h || h->W <= 0) return 0;
  if (!h->children.empty()) return uint64_t(h->W) * h->H;
  return h->world <= 1 ? uint64_t(h->W) * h->H : h->outputPixels();
}

// A frame of a multi-device handle: every device marches its tiles into the same destination frame on its own stream;
// the caller's stream waits for all of them (async) or the host does (synchronous).
static int renderMulti(ExaHipRenderer *h, uint32_t *rgba8, int32_t dstIsDevice, hipStream_t s, bool async, bool stats)
{
  if (h->W <= 0) { h->fail("exa_hip_render: framebuffer not sized"); return 1; }
  uint32_t *dst = dstIsDevice && rgba8 ? rgba8 : h->color.p;
  const bool willSync = !(async && dstIsDevice && !stats);
  {
    EXA_ON_DEVICE(h);
    HIP_TRY(h, hipEventRecord(h->evCall, s));           // what the caller queued before (e.g. the copy-out of this buffer)
  }
  for (ExaHipRenderer *c : h->children) {
    DeviceGuard g(c->device);
    if (g.err != hipSuccess) { h->fail("hipSetDevice failed"); return 1; }
    HIP_TRY(h, hipStreamWaitEvent(c->ownStream, h->evCall, 0));
    if (c->prepareFrame(c->ownStream)) { h->fail(c->err); return 1; }
    if (stats) HIP_TRY(h, hipMemsetAsync(c->statsBuf.p, 0, ST_COUNT * sizeof(unsigned long long), c->ownStream));
    c->measureCosts = willSync && !stats;
    if (c->launch(dst, stats, c->ownStream)) { h->fail(c->err); return 1; }
  }
  if (!willSync) {
    EXA_ON_DEVICE(h);
    for (ExaHipRenderer *c : h->children) HIP_TRY(h, hipStreamWaitEvent(s, c->ev1, 0));
    return 0;
  }
  ExaHipStats sum{};
  for (ExaHipRenderer *c : h->children) {
    DeviceGuard g(c->device);
    HIP_TRY(h, hipEventSynchronize(c->ev1));
    HIP_TRY(h, hipEventElapsedTime(&c->last.kernel_ms, c->ev0, c->ev1));
    if (c->measureCosts && c->feedback && c->costPhase == 1 && c->useKd() && c->reorderFromCosts()) { h->fail(c->err); return 1; }
    int32_t flag = 0;
    HIP_TRY(h, hipMemcpy(&flag, c->errorFlag.p, sizeof(flag), hipMemcpyDeviceToHost));
    if (flag) {
      (void)hipMemset(c->errorFlag.p, 0, sizeof(int32_t));
      h->fail("exa_hip_render: a ray-march loop guard tripped (step size too small for the ray length?)");
      return 3;
    }
    sum.kernel_ms = std::max(sum.kernel_ms, c->last.kernel_ms);
    sum.rebuild_ms = std::max(sum.rebuild_ms, c->last.rebuild_ms);
    sum.node_bytes = c->last.node_bytes;
    if (stats) {
      unsigned long long k[ST_COUNT];
      HIP_TRY(h, hipMemcpy(k, c->statsBuf.p, sizeof(k), hipMemcpyDeviceToHost));
      sum.segments += k[ST_SEGMENTS]; sum.sample_evals += k[ST_SAMPLE_EVALS]; sum.samples += k[ST_SAMPLES];
      sum.brick_visits += k[ST_BRICK_VISITS]; sum.corner_loads += k[ST_CORNER_LOADS];
      sum.iso_segments += k[ST_ISO_SEGMENTS]; sum.iso_evals += k[ST_ISO_EVALS]; sum.nodes_visited += k[ST_NODES];
      for (int i = 0; i < 9; i++) sum.diag[i] += k[ST_W_BRICK + i];
      for (int i = 0; i < 5; i++) sum.phase_cycles[i] += k[ST_T_BRICK + i];
      sum.walk_restarts += k[ST_RESTARTS]; sum.walk_union_nodes += k[ST_UNION]; sum.walk_probe_overflow += k[ST_PROBE_OVERFLOW];
      sum.wave_iters += k[ST_WAVE_ITERS]; sum.tile_iters += k[ST_TILE_ITERS]; sum.walk_leaf_visits += k[ST_ROPE_LEAVES];
    }
  }
  sum.pixels = uint64_t(h->W) * h->H;
  if (!stats) {                                  // keep the counters of the last counted frame, as a single-device handle does
    const ExaHipStats keep = h->last;
    sum.segments = keep.segments; sum.sample_evals = keep.sample_evals; sum.samples = keep.samples; sum.brick_visits = keep.brick_visits;
    sum.corner_loads = keep.corner_loads; sum.iso_segments = keep.iso_segments; sum.iso_evals = keep.iso_evals; sum.nodes_visited = keep.nodes_visited;
    for (int i = 0; i < 9; i++) sum.diag[i] = keep.diag[i];
    for (int i = 0; i < 5; i++) sum.phase_cycles[i] = keep.phase_cycles[i];
    sum.walk_restarts = keep.walk_restarts; sum.walk_union_nodes = keep.walk_union_nodes; sum.walk_probe_overflow = keep.walk_probe_overflow;
    sum.wave_iters = keep.wave_iters; sum.tile_iters = keep.tile_iters; sum.walk_leaf_visits = keep.walk_leaf_visits;
  }
  h->last = sum;
  if (!dstIsDevice && rgba8) {
    EXA_ON_DEVICE(h);
    HIP_TRY(h, hipMemcpy(rgba8, h->color.p, size_t(h->W) * h->H * sizeof(uint32_t), hipMemcpyDeviceToHost));
  }
  return 0;
}

static int renderImpl(ExaHipRenderer *h, uint32_t *rgba8, int32_t dstIsDevice, hipStream_t s, bool async, bool stats)
{
  if (!h) return 1;
  if (!h->children.empty()) return renderMulti(h, rgba8, dstIsDevice, s, async, stats);
  EXA_ON_DEVICE(h);
  if (h->prepareFrame(s)) return 1;
  uint32_t *dst = dstIsDevice && rgba8 ? rgba8 : h->color.p;
  if (stats) HIP_TRY(h, hipMemsetAsync(h->statsBuf.p, 0, ST_COUNT * sizeof(unsigned long long), s));
  const bool willSync = !(async && dstIsDevice && !stats);
  h->measureCosts = willSync && !stats;
  if (h->launch(dst, stats, s)) return 1;
  if (!willSync) return 0;
  HIP_TRY(h, hipEventSynchronize(h->ev1));
  HIP_TRY(h, hipEventElapsedTime(&h->last.kernel_ms, h->ev0, h->ev1));
  if (h->measureCosts && h->feedback && h->costPhase == 1 && h->useKd() && h->reorderFromCosts()) return 1;
  const size_t px = (size_t)exa_hip_output_pixels(h);
  if (!dstIsDevice && rgba8) HIP_TRY(h, hipMemcpy(rgba8, h->color.p, px * sizeof(uint32_t), hipMemcpyDeviceToHost));
  int32_t flag = 0;
  HIP_TRY(h, hipMemcpy(&flag, h->errorFlag.p, sizeof(flag), hipMemcpyDeviceToHost));
  if (flag) {
    (void)hipMemset(h->errorFlag.p, 0, sizeof(int32_t));
    h->fail("exa_hip_render: a ray-march loop guard tripped (step size too small for the ray length?)");
    return 3;
  }
  if (stats) {
    unsigned long long c[ST_COUNT];
    HIP_TRY(h, hipMemcpy(c, h->statsBuf.p, sizeof(c), hipMemcpyDeviceToHost));
    h->last.segments = c[ST_SEGMENTS]; h->last.sample_evals = c[ST_SAMPLE_EVALS]; h->last.samples = c[ST_SAMPLES];
    h->last.brick_visits = c[ST_BRICK_VISITS]; h->last.corner_loads = c[ST_CORNER_LOADS];
    h->last.iso_segments = c[ST_ISO_SEGMENTS]; h->last.iso_evals = c[ST_ISO_EVALS]; h->last.nodes_visited = c[ST_NODES];
    for (int i = 0; i < 9; i++) h->last.diag[i] = c[ST_W_BRICK + i];
    for (int i = 0; i < 5; i++) h->last.phase_cycles[i] = c[ST_T_BRICK + i];
    h->last.walk_restarts = c[ST_RESTARTS]; h->last.walk_union_nodes = c[ST_UNION]; h->last.walk_probe_overflow = c[ST_PROBE_OVERFLOW];
    h->last.wave_iters = c[ST_WAVE_ITERS]; h->last.tile_iters = c[ST_TILE_ITERS]; h->last.walk_leaf_visits = c[ST_ROPE_LEAVES];
    h->walkProbe.release();
  }
  h->last.pixels = px;
  return 0;
}

int exa_hip_render(ExaHipRenderer *h, uint32_t *rgba8, int32_t dstIsDevice, void *hipStream, int32_t async)
{ return renderImpl(h, rgba8, dstIsDevice, (hipStream_t)hipStream, async != 0, false); }

int exa_hip_render_stats(ExaHipRenderer *h, uint32_t *rgba8, int32_t dstIsDevice, ExaHipStats *out)
{
  const int rc = renderImpl(h, rgba8, dstIsDevice, nullptr, false, true);
  if (rc == 0 && out) *out = h->last;
  return rc;
}

int exa_hip_get_stats(ExaHipRenderer *h, ExaHipStats *out)
{
  if (!h || !out) return 1;
  if (!h->children.empty()) {                  // kernel time of the slowest device (refreshed if an async frame has completed)
    float ms = 0.f;
    for (ExaHipRenderer *c : h->children) { ExaHipStats s; exa_hip_get_stats(c, &s); ms = std::max(ms, s.kernel_ms); }
    h->last.kernel_ms = ms;
    *out = h->last;
    return 0;
  }
  // refresh the kernel time of an async launch if it has completed
  if (h->ev0 && hipEventQuery(h->ev1) == hipSuccess) (void)hipEventElapsedTime(&h->last.kernel_ms, h->ev0, h->ev1);
  *out = h->last;
  return 0;
}

int exa_hip_untile(ExaHipRenderer *h, const uint32_t *gathered, uint64_t shardStridePixels,
                   int32_t worldSize, uint32_t *rgba8_out, void *hipStream)
{
  if (!h || !gathered || !rgba8_out || worldSize < 1) return 1;
  EXA_ON_DEVICE(h);
  HIP_TRY(h, launchUntile(gathered, shardStridePixels, worldSize, h->W, h->H, rgba8_out, (hipStream_t)hipStream));
  return 0;
}

// row-major frame <-> the tile-major shards of a multi-device handle's children (host side)
static int multiAccum(ExaHipRenderer *h, float *frame4, bool read)
{
  const int n = (int)h->children.size(), W = h->W, H = h->H;
  const int tilesX = (W + kTile - 1) / kTile, tilesY = (H + kTile - 1) / kTile;
  for (int i = 0; i < n; i++) {
    ExaHipRenderer *c = h->children[i];
    if (c->accum.n == 0) continue;                 // more devices than tiles: this one owns nothing
    std::vector<float> shard(c->accum.n * 4);
    if (read && exa_hip_read_accum(c, shard.data())) { h->fail(c->err); return 1; }
    if (!read && exa_hip_read_accum(c, shard.data())) { h->fail(c->err); return 1; }   // keep the padding pixels of ragged tiles
    for (int t = i; t < tilesX * tilesY; t += n) {
      const int tx = t % tilesX, ty = t / tilesX;
      for (int y = 0; y < kTile && ty * kTile + y < H; y++)
        for (int x = 0; x < kTile && tx * kTile + x < W; x++) {
          float *f = frame4 + 4 * (size_t(tx * kTile + x) + size_t(W) * (ty * kTile + y));
          float *s = shard.data() + 4 * (size_t(t / n) * kTilePixels + size_t(y) * kTile + x);
          for (int k = 0; k < 4; k++) { if (read) f[k] = s[k]; else s[k] = f[k]; }
        }
    }
    if (!read && exa_hip_write_accum(c, shard.data())) { h->fail(c->err); return 1; }
  }
  return 0;
}

int exa_hip_read_accum(ExaHipRenderer *h, float *dst4)
{
  if (!h || !dst4) return 1;
  if (!h->children.empty()) return multiAccum(h, dst4, true);
  EXA_ON_DEVICE(h);
  HIP_TRY(h, hipMemcpy(dst4, h->accum.p, h->accum.n * sizeof(float4), hipMemcpyDeviceToHost));
  return 0;
}

int exa_hip_write_accum(ExaHipRenderer *h, const float *src4)
{
  if (!h || !src4) return 1;
  if (!h->children.empty()) return multiAccum(h, const_cast<float *>(src4), false);
  EXA_ON_DEVICE(h);
  HIP_TRY(h, hipMemcpy(h->accum.p, src4, h->accum.n * sizeof(float4), hipMemcpyHostToDevice));
  return 0;
}

int exa_hip_read_activity(ExaHipRenderer *h, int32_t which, uint8_t *dst)
{
  if (h && !h->children.empty()) {
    const int rc = exa_hip_read_activity(h->children[0], which, dst);
    if (rc) h->fail(h->children[0]->err);
    return rc;
  }
  if (!h || !dst) return 1;
  EXA_ON_DEVICE(h);
  if (h->prepareFrame(nullptr)) return 1;
  if (which == 1 && !h->isoEnabled()) {     // evaluate on demand
    HIP_TRY(h, launchIsoActivity(h->sc, h->fs, h->isoActive.p, nullptr));
  }
  HIP_TRY(h, hipDeviceSynchronize());
  HIP_TRY(h, hipMemcpy(dst, which ? h->isoActive.p : h->volActive.p, h->sc.numRegions, hipMemcpyDeviceToHost));
  return 0;
}

} // extern "C"
