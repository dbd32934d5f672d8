// exa_lbvh.hip — device-side build of the LBVH over the brick regions (north_star: "a software LBVH over the
// brick-regions replaces the OptiX BVH"; the reference builds its BVH inside OptiX, exa/OptixRenderer.cpp:614-721).
//
// Same tree as the host builder in exa_module.cpp (LbvhTopology), node for node and id for id:
//   1. 63-bit Morton code of every region's box centre (21 bits per axis over the union box, double arithmetic)
//   2. stable radix sort of (code, region)                      — hipCUB/rocPRIM, the only library call on this path
//   3. topology level by level from the root: a node over the sorted range [lo, hi) splits after the last code
//      that shares more leading bits with the first one than the last one does (Karras' rule), unless that would
//      leave a side too large for the remaining depth budget (the traversal stack holds kStackDepth entries), in
//      which case it splits at the median.  A node's id is its preorder number, which is lo + (left turns on the
//      path from the root): known when the node is created, so the levels can be built in parallel.
// Boxes are filled by the refit (exa_kernels.hip), deepest level first.
#include "exa_device.h"

#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <vector>

namespace exa {

namespace {

// order-preserving map float -> uint32 for atomicMin/atomicMax
__device__ __forceinline__ uint32_t orderedBits(float f)
{
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float fromOrderedBits(uint32_t u)
{
  const uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  float f;
#ifdef __HIP_DEVICE_COMPILE__
  f = __uint_as_float(b);
#else
  memcpy(&f, &b, sizeof(f));
#endif
  return f;
}

// minmax[0..2] = min of box lower (ordered bits), minmax[3..5] = max of box upper
__global__ __launch_bounds__(256) void lbvhBoundsKernel(const float *boxes, uint32_t n, uint32_t *minmax)
{
  float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
    for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], boxes[6 * size_t(i) + k]); hi[k] = fmaxf(hi[k], boxes[6 * size_t(i) + 3 + k]); }
  for (int k = 0; k < 3; k++) {
    for (int off = 32; off > 0; off >>= 1) {
      lo[k] = fminf(lo[k], __shfl_down(lo[k], off, 64));
      hi[k] = fmaxf(hi[k], __shfl_down(hi[k], off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&minmax[k], orderedBits(lo[k]));
      atomicMax(&minmax[3 + k], orderedBits(hi[k]));
    }
  }
}

__device__ __forceinline__ unsigned long long spread21(unsigned long long v)
{
  v &= 0x1fffffull;
  v = (v | v << 32) & 0x1f00000000ffffull;
  v = (v | v << 16) & 0x1f0000ff0000ffull;
  v = (v | v << 8) & 0x100f00f00f00f00full;
  v = (v | v << 4) & 0x10c30c30c30c30c3ull;
  v = (v | v << 2) & 0x1249249249249249ull;
  return v;
}

__global__ __launch_bounds__(256) void lbvhCodesKernel(const float *boxes, uint32_t n, const uint32_t *minmax,
                                                       unsigned long long *codes, uint32_t *idx)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  unsigned long long code = 0;
  for (int k = 0; k < 3; k++) {
    const double lo = (double)fromOrderedBits(minmax[k]), hi = (double)fromOrderedBits(minmax[3 + k]);
    const double c = 0.5 * ((double)boxes[6 * size_t(i) + k] + (double)boxes[6 * size_t(i) + 3 + k]);
    const double ext = hi - lo;
    double u = ext > 0 ? (c - lo) / ext : 0.0;
    u = fmin(fmax(u, 0.0), 1.0);
    unsigned long long q = (unsigned long long)(u * 2097152.0);
    q = q < 2097151ull ? q : 2097151ull;
    code |= spread21(q) << k;
  }
  codes[i] = code;
  idx[i] = i;
}

struct LbvhWork { uint32_t lo, hi, lefts; int32_t parent; uint32_t side; };

__device__ __forceinline__ int ceilLog2(unsigned long long n) { int l = 0; while ((1ull << l) < n) l++; return l; }

__global__ __launch_bounds__(256) void lbvhLevelKernel(const LbvhWork *cur, uint32_t count, int depth,
                                                       const unsigned long long *codes, const uint32_t *order,
                                                       BvhNode *nodes, LbvhWork *next, uint32_t *nextCount,
                                                       int32_t *levelIds, uint32_t *levelCount)
{
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= count) return;
  const LbvhWork w = cur[j];
  const uint32_t n = w.hi - w.lo;
  int32_t ref;
  if (n == 1) {
    ref = ~int32_t(order[w.lo]);
  } else {
    const int32_t me = int32_t(w.lo + w.lefts);            // preorder number
    ref = me;
    uint32_t split = w.lo + (n + 1) / 2;                   // median fallback
    const unsigned long long first = codes[w.lo], last = codes[w.hi - 1];
    if (first != last) {
      const int prefix = __clzll((long long)(first ^ last));
      uint32_t at = w.lo, step = w.hi - 1 - w.lo;
      do {
        step = (step + 1) >> 1;
        const uint32_t cand = at + step;
        if (cand < w.hi - 1) {
          const unsigned long long x = first ^ codes[cand];
          const int pfx = x ? __clzll((long long)x) : 64;
          if (pfx > prefix) at = cand;
        }
      } while (step > 1);
      const uint32_t s = at + 1;
      const uint32_t big = max(s - w.lo, w.hi - s);
      if (ceilLog2(big) <= kStackDepth - 1 - depth) split = s;   // keep the internal depth <= kStackDepth-1
    }
    BvhNode nd;
    nd.q0 = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX);     // both child boxes empty until the first refit
    nd.q1 = make_float4(-FLT_MAX, -FLT_MAX, FLT_MAX, FLT_MAX);
    nd.q2 = make_float4(FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
    nd.child0 = 0; nd.child1 = 0; nd.pad0 = nd.pad1 = 0;
    // the children fill child0/child1 in the next launch; the box part is written here
    float4 *np = reinterpret_cast<float4 *>(nodes + me);
    np[0] = nd.q0; np[1] = nd.q1; np[2] = nd.q2;
    nodes[me].pad0 = 0; nodes[me].pad1 = 0;
    const uint32_t at2 = atomicAdd(nextCount, 2u);
    LbvhWork a; a.lo = w.lo; a.hi = split; a.lefts = w.lefts + 1; a.parent = me; a.side = 0;
    LbvhWork b; b.lo = split; b.hi = w.hi; b.lefts = w.lefts; b.parent = me; b.side = 1;
    next[at2] = a; next[at2 + 1] = b;
    levelIds[atomicAdd(levelCount, 1u)] = me;
  }
  if (w.parent >= 0) {
    if (w.side) nodes[w.parent].child1 = ref; else nodes[w.parent].child0 = ref;
  }
}

} // namespace

#define LB_TRY(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return e_; } while (0)

// boxes: numPrims x 6 floats (lo, hi) on the device, numPrims >= 2.  Fills nodes[numPrims-1] (children; boxes empty)
// and levelIds[numPrims-1] = the internal node ids grouped by depth; levelCounts (host) = internal nodes per depth.
hipError_t buildLbvhTopologyDevice(const float *boxes, uint32_t numPrims, BvhNode *nodes, int32_t *levelIds,
                                   std::vector<uint32_t> &levelCounts, hipStream_t s)
{
  levelCounts.clear();
  const uint32_t n = numPrims;
  uint32_t *minmax = nullptr, *idxIn = nullptr, *idxOut = nullptr, *counters = nullptr;
  unsigned long long *codesIn = nullptr, *codesOut = nullptr;
  LbvhWork *workA = nullptr, *workB = nullptr;
  void *tmp = nullptr;
  size_t tmpBytes = 0;
  auto release = [&]() {
    (void)hipFree(minmax); (void)hipFree(idxIn); (void)hipFree(idxOut); (void)hipFree(counters);
    (void)hipFree(codesIn); (void)hipFree(codesOut); (void)hipFree(workA); (void)hipFree(workB); (void)hipFree(tmp);
  };
#define LB_GO(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { release(); return e_; } } while (0)
  LB_GO(hipMalloc((void **)&minmax, 6 * sizeof(uint32_t)));
  LB_GO(hipMalloc((void **)&idxIn, size_t(n) * sizeof(uint32_t)));
  LB_GO(hipMalloc((void **)&idxOut, size_t(n) * sizeof(uint32_t)));
  LB_GO(hipMalloc((void **)&codesIn, size_t(n) * sizeof(unsigned long long)));
  LB_GO(hipMalloc((void **)&codesOut, size_t(n) * sizeof(unsigned long long)));
  LB_GO(hipMalloc((void **)&counters, 2 * sizeof(uint32_t)));
  // a level holds at most n work items (the ranges of one level are disjoint)
  LB_GO(hipMalloc((void **)&workA, size_t(n) * sizeof(LbvhWork)));
  LB_GO(hipMalloc((void **)&workB, size_t(n) * sizeof(LbvhWork)));
  {
    const uint32_t init[6] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u };
    LB_GO(hipMemcpyAsync(minmax, init, sizeof(init), hipMemcpyHostToDevice, s));
  }
  const uint32_t blocks = (n + 255u) / 256u;
  hipLaunchKernelGGL(lbvhBoundsKernel, dim3(blocks < 4096u ? blocks : 4096u), dim3(256), 0, s, boxes, n, minmax);
  hipLaunchKernelGGL(lbvhCodesKernel, dim3(blocks), dim3(256), 0, s, boxes, n, minmax, codesIn, idxIn);
  LB_GO(hipGetLastError());
  LB_GO(hipcub::DeviceRadixSort::SortPairs(nullptr, tmpBytes, codesIn, codesOut, idxIn, idxOut, (int)n, 0, 63, s));
  LB_GO(hipMalloc(&tmp, tmpBytes ? tmpBytes : 16));
  LB_GO(hipcub::DeviceRadixSort::SortPairs(tmp, tmpBytes, codesIn, codesOut, idxIn, idxOut, (int)n, 0, 63, s));
  // level-by-level topology
  LbvhWork root; root.lo = 0; root.hi = n; root.lefts = 0; root.parent = -1; root.side = 0;
  LB_GO(hipMemcpyAsync(workA, &root, sizeof(root), hipMemcpyHostToDevice, s));
  uint32_t count = 1, idsDone = 0;
  LbvhWork *cur = workA, *nxt = workB;
  for (int depth = 0; count > 0; depth++) {
    if (depth > 2 * kStackDepth) { release(); return hipErrorUnknown; }      // cannot happen: the depth budget forces medians
    LB_GO(hipMemsetAsync(counters, 0, 2 * sizeof(uint32_t), s));
    hipLaunchKernelGGL(lbvhLevelKernel, dim3((count + 255u) / 256u), dim3(256), 0, s, cur, count, depth, codesOut, idxOut,
                       nodes, nxt, counters, levelIds + idsDone, counters + 1);
    LB_GO(hipGetLastError());
    uint32_t c[2];
    LB_GO(hipMemcpyAsync(c, counters, sizeof(c), hipMemcpyDeviceToHost, s));
    LB_GO(hipStreamSynchronize(s));
    if (c[1]) levelCounts.push_back(c[1]);
    idsDone += c[1];
    count = c[0];
    LbvhWork *t = cur; cur = nxt; nxt = t;
  }
#undef LB_GO
  release();
  return idsDone == n - 1 ? hipSuccess : hipErrorUnknown;
}

} // namespace exa
