// exa_device.h — device-side data layout shared by the module (exa_module.cpp)
// and the kernels (exa_kernels.hip).  gfx950 only.
#pragma once
#include "../../include/exa_hip.h"
#include <hip/hip_runtime.h>

#include <vector>

// ---- build-time constants shared by the module and the kernels ----
#ifndef EXA_MARCH_WAVES
#define EXA_MARCH_WAVES 6      // waves per SIMD the one-channel march is compiled for (80 VGPRs, 26 KB of LDS per workgroup with the stack walk) ...
#endif
#ifndef EXA_ROPE_WAVES
#define EXA_ROPE_WAVES 7       // ... and its rope-walk variant in the default association of the basis sums (72 VGPRs, 22 KB of LDS
                               // per workgroup; exa_kernels.hip: marchWaves, LEAN)
#endif
#ifndef EXA_MULTI_WAVES
#define EXA_MULTI_WAVES 6      // ... and the multi-channel march (80 VGPRs; two TF tables + a 3-entry stack: 25 KB of LDS per workgroup)
#endif
#ifndef EXA_PREPASS_WAVES
#define EXA_PREPASS_WAVES 4    // waves per SIMD the surfaces pre-pass is compiled for: 2/3/4/5/6 -> 20.2/15.1/12.6/12.6/16.9 ms on C5
#endif
#ifndef EXA_IL2_WAVES
#define EXA_IL2_WAVES 5        // waves per SIMD of the channel-interleaved march with two channels ...
#endif
#ifndef EXA_IL34_WAVES
#define EXA_IL34_WAVES 4       // ... and with three or four (their cell values and sums need the registers)
#endif
#ifndef EXA_PREPASS_ISO_WAVES
#define EXA_PREPASS_ISO_WAVES 4 // ... and its variant for frames whose only surfaces are implicit iso-surfaces (125 VGPRs, no scratch).
                                // C3 is bound by the latency of its longest rays there (3 / 4 / 5 waves: 21.08 / 21.07 / 21.62 ms per frame),
                                // C5 by throughput (3 waves + the march header kept in registers across a one-brick segment: 13.9 instead of 10.9 ms)
#endif
#ifndef EXA_AO_ISO_WAVES
#define EXA_AO_ISO_WAVES 6      // waves per SIMD of the deferred AO rays' kernel (iso-only frames): the rays are latency-bound marches
#endif
#ifndef EXA_EMPTY_CELLS
#define EXA_EMPTY_CELLS 0     // this translation unit of exa_kernels.hip skips corners whose cell holds the poison value
#endif
#ifndef EXA_BASIS_FORM
#define EXA_BASIS_FORM 0      // which association of the basis sums this translation unit of exa_kernels.hip is compiled for
#endif
namespace exa {

// One LBVH node = 64 bytes = four 16-byte loads.  Both children's boxes live in
// the parent, so a leaf is tested without touching its region record:
//   q0 = (lo0.x lo0.y lo0.z hi0.x)  q1 = (hi0.y hi0.z lo1.x lo1.y)
//   q2 = (lo1.z hi1.x hi1.y hi1.z)  c  = (child0, child1, -, -)
// child >= 0: internal node index; child < 0: leaf, region id = ~child.
// A child whose subtree holds no active region has lo.x > hi.x (refit writes
// lo = +FLT_MAX, hi = -FLT_MAX) and is skipped.
struct alignas(64) BvhNode {
  float4 q0, q1, q2;
  int32_t child0, child1, pad0, pad1;
};
static_assert(sizeof(BvhNode) == 64, "node must be one 64-byte line");

// what the march needs from a region (16 B, one load per segment); the box comes
// from the LBVH node, the value range is only needed by the activity kernel.
struct alignas(16) RegionInfo {
  int32_t listBegin;      // offset into leafList
  int32_t listSize;       // k = number of bricks overlapping in this region
  float   finestLevelCellWidth;
  int32_t firstBrick;     // leafList[listBegin], saves a dependent load
};

// kd node of the region partition (16 B, one load): the split plane and, refreshed by
// the refit pass, which children hold an active region.
//   word: bits 0-1 axis | bit 2 left active (volume) | bit 3 right active (volume)
//                       | bit 4 left active (iso)    | bit 5 right active (iso)
struct alignas(16) KdNodeDev {
  float    split;
  uint32_t word;
  int32_t  left, right;     // >= 0 node, < 0 leaf (~region), EXA_KD_EMPTY nothing
};
static_assert(sizeof(KdNodeDev) == 16, "kd node = one 16-byte load");

// Leaf of the rope walk (64 B = four 16-byte loads), one per region plus one per gap (a child slot of the region kd-tree
// that holds nothing: space no brick covers): its box, what the march needs from the region, whether it is active, and one
// link per face to whatever lies across it — an inner node of ropeNodes (>= 0), a leaf (~index < 0), or EXA_KD_DONE
// outside the root box.  The walk goes from leaf to leaf through these links: no stack, no restart from the root.
struct alignas(16) RopeLeaf {
  float    lo[3], hi0;      // domain lower, domain upper.x
  float    hi1, hi2;        // domain upper.y, .z
  uint32_t rec;             // the march tree's leaf reference: packed {listBegin | listSize-1 | level} (RenderArgs::leafBeginBits)
  uint32_t flags;           // bit 0 active for the volume march, bit 1 active for the iso march (refreshed with the activity)
  int32_t  rope[6];         // -x +x -y +y -z +z
  int32_t  region;          // region id, -1 for a gap
  int32_t  pad;
};
static_assert(sizeof(RopeLeaf) == 64, "rope leaf = four 16-byte loads");

// region record of the kd path: exact domain for the slab test plus the march data (48 B)
struct alignas(16) RegionRec {
  float   lo[3], hi0;       // domain lower, domain upper.x
  float   hi1, hi2, finestLevelCellWidth; int32_t firstBrick;
  int32_t listBegin, listSize, pad0, pad1;
};
static_assert(sizeof(RegionRec) == 48, "region record = three 16-byte loads");

struct DeviceScene {
  const int4       *bricks;        // ExaBrick as two int4: (lower.xyz,size.x) (size.yz,level,begin)
  const int32_t    *leafList;
  const int4       *leafHdr;       // march headers along the leaf list, two int4 per entry:
                                   // (float(lower.xyz), 2^-level | size.xyz, begin), see exa_module.cpp
  const float      *scalars;
  const RegionInfo *regionInfo;
  const float2     *valueRange;    // per region
  const float      *domain;        // per region 6 floats (lo, hi) for refit
  unsigned long long channelOffset[EXA_MAX_CHANNELS];
  uint32_t numRegions;
  uint32_t numInternal;            // internal LBVH nodes
};

#ifndef EXA_KD_STACK
#define EXA_KD_STACK 4      // per-lane short stack of the kd walk (LDS, 12 B per entry) ...
#endif
#ifndef EXA_KD_STACK_MULTI
#define EXA_KD_STACK_MULTI 3   // the multi-channel march: one entry fewer, so that two TF tables + stack + queue fit 6 workgroups per CU
#endif
#ifndef EXA_ROPE_QUEUE
#define EXA_ROPE_QUEUE 5    // segment queue of the rope walk: it keeps no stack, so the queue gets the stack's LDS as well — 16-byte
                            // entries {region record, t1, first sample's t_i, t0}.  Five of them (80 B per lane + one TF table = 22 KB
                            // per workgroup) let seven workgroups share a CU; six entries at six waves measure the same as five
#endif
#ifndef EXA_ROPE_QUEUE_MULTI
#define EXA_ROPE_QUEUE_MULTI 5   // ... with two TF tables in LDS (the multi-channel march): 80 B per lane keep six workgroups per CU
#endif
#ifndef EXA_SEG_QUEUE
#define EXA_SEG_QUEUE 4     // ... and per-lane queue of accepted segments (12 B per entry): 96 B per lane = 6 workgroups per CU.
                            // Measured on C4 (stack/queue), bursts that end when no lane is dry: 4/4 22.42 ms, 3/5 22.56,
                            // 5/3 22.89, 2/6 23.41 (with bursts run until every queue is full: 4/4 24.62, 3/5 24.25)
#endif
// entries of the short stack in the LDS the 12-byte layout reserves (kKdStack / kKdStackMulti x 12 bytes per lane)
enum { kKdStackEntries = EXA_KD_STACK, kKdStackMultiEntries = EXA_KD_STACK_MULTI };
enum { kTile = 16, kTilePixels = 256, kStackDepth = 32, kKdStack = EXA_KD_STACK, kKdStackMulti = EXA_KD_STACK_MULTI, kSegQueue = EXA_SEG_QUEUE,
       kRopeQueue = EXA_ROPE_QUEUE, kRopeQueueMulti = EXA_ROPE_QUEUE_MULTI,
       kKdBlock = 256,        // threads per workgroup of the kd kernel (measured on C4: 256 -> 38.1 ms, 128 -> 41.5, 64 -> 42.7)
       kWideSegCap = 256,
       kWalkProbeBits = 15, kWalkProbeSize = 1 << kWalkProbeBits };   // walk probe: entries of a wave's node set   // wide march: leaves a window walker lists per round (16 B each; a fuller window takes more rounds)

enum StatSlot { ST_SEGMENTS, ST_SAMPLE_EVALS, ST_SAMPLES, ST_BRICK_VISITS, ST_CORNER_LOADS,
                ST_ISO_SEGMENTS, ST_ISO_EVALS, ST_NODES,
                // wave-level diagnostics of the instrumented v2 kernel: executions of a phase by a wave,
                // and lanes active in them (lane utilisation = lanes / (64 * waves))
                ST_W_BRICK, ST_L_BRICK, ST_W_FINAL, ST_L_FINAL, ST_W_NODE, ST_L_NODE, ST_W_LEAF, ST_L_LEAF,
                ST_KD_MISMATCH,
                // shader-clock cycles of the waves by phase (sum over waves): brick visit, sample epilogue, kd walk,
                // segment pop, everything else (ray set-up, output)
                ST_T_BRICK, ST_T_FINAL, ST_T_WALK, ST_T_SEG, ST_T_OTHER,
                ST_RESTARTS, ST_UNION, ST_PROBE_OVERFLOW, ST_WAVE_ITERS, ST_TILE_ITERS,
                ST_ROPE_LEAVES,         // rope walk: leaves fetched (each counts as four 16-byte nodes in ST_NODES)
                ST_COUNT };

// a shaded surface hit whose ambient-occlusion rays aoRaysKdKernel traces: hit point + |cos| of the primary ray, normal +
// ambient term, base colour + the LCG state its two samples draw from, pixel slot
struct alignas(16) AoRecord { float4 posFd, ngAmb, baseRnd; uint32_t slot, pad0, pad1, pad2; };
static_assert(sizeof(AoRecord) == 64, "AO record = four 16-byte stores");

struct RenderArgs {
  DeviceScene        sc;
  const BvhNode     *volNodes;
  const BvhNode     *isoNodes;
  const KdNodeDev   *kdNodes;       // region kd-tree (NULL: LBVH only)
  const KdNodeDev   *kdMarchNodes;  // the tree the volume march walks: kdNodes, or its copy whose leaf references are
                                    // the packed region records {listBegin | listSize-1 | level} (leafBeginBits != 0)
  int32_t            kdMarchRoot;
  uint32_t           leafBeginBits, leafSizeBits;
  const RegionRec   *regionRec;
  // rope walk of the DVR march (NULL: the stack walk): leaves with neighbour links + the tree to descend in behind a link
  const RopeLeaf    *ropeLeaves;
  const KdNodeDev   *ropeNodes;
  int32_t            ropeRoot;
  int32_t            ropeFastDiv;   // the scene's planes allow the short exact division (see ropeStep)
  int32_t            ropeAddr32;    // ... and its leaf / node arrays are below 4 GiB
  int32_t            kdRoot;
  int32_t            kdIsoRoot;     // where the iso walk starts: kdRoot, or EXA_KD_EMPTY + 1 (= done) when the tree is one inactive leaf
  float              kdLo[3], kdHi[3];   // box of the kd root = union of all brick domains
  const BvhNode     *meshNodes;     // BVH over the triangle surfaces (NULL: none)
  const float       *meshVerts;     // 3 floats per vertex, world space
  const int32_t     *meshTris;      // 3 indices per triangle
  int32_t            numTris;
  const BvhNode     *streamNodes;   // BVH over the visible streamline segments (NULL: none)
  const float       *traces;        // numTraces x numTimesteps x 3
  int32_t            numStreamPrims;
  int32_t            tracerChannels[3], numTraces, numTimesteps, timestep;
  float              steplen;
  float              worldLo[3], worldHi[3];   // worldSpaceBounds (OptixRenderer.cpp:330-332), contour planes
  ExaHipFrameState   fs;
  ExaHipParams       p;
  const float4      *xf;            // numXfChannels x 128 (r,g,b,a)
  const float       *cellsIl;       // channel-interleaved copy of the primary channels, float[cell][numPrimaryChannels]
                                    // (NULL: none; the march then reads the fields one after the other)
  int32_t            il32;          // ... and it is below 4 GiB (32-bit byte offsets)
  int32_t            mul24;         // every brick: sizes < 2^24, size.x*size.y < 2^24, cells < 2^32 -> 24-bit multiplies
  int32_t            addr32;        // a scalar field, the march headers and the kd nodes are each below 4 GiB ->
                                    // 32-bit byte offsets from a uniform base
  float              invDtPow2;     // 1/dt when launch.dt is a power of two (exact), else 0
  int32_t            fastSampler;   // surfaces pre-pass: samplePoint on the march headers with the masked-weight basis
  float              tfFracMagic;   // 2^15: TF filter weight rounded to 8 fractional bits (CUDA tex1D), 0: full precision
  int32_t            numXfChannels;
  int32_t            W, H, tilesX, tilesY;
  int32_t            rank, world;   // image-space shard
  const int32_t     *tileMap;       // blockIdx.x -> global tile id (launch order)
  uint32_t          *color;
  int32_t            colorRowMajor; // sharded handle whose colour goes row-major into a full frame (multi-device handle:
                                    // a peer-mapped pointer to the root device's buffer); accum / surf stay tile-major
  float4            *accum;
  float4            *surf;          // surfaces pre-pass -> march: {background rgb, surface t_hit} per pixel slot
  float4            *pixOut;        // != null: the march stores its pixel colour here instead of finishing the pixel (compositeKdKernel does)
  uint32_t          *surfRnd;       // LCG state after the pre-pass' draws
  unsigned long long *stats;        // ST_COUNT counters (instrumented variant)
  int32_t           *errorFlag;     // set when a loop guard trips
  int32_t            debugPixel;    // >= 0: only pixel x + W*y is rendered (debugging aid)
  const int32_t     *wideTileMap;   // wide march: launch slot / L -> global tile id
  float4            *wideSegs;      // wide march: [tile of this launch][ray][window][kWideSegCap] {record, tn, tf, -}
  uint32_t          *walkProbe;     // != null (counting variant, option walk_probe): per wave a hash set of kWalkProbeSize node ids
  AoRecord          *aoRecs;        // surfaces pre-pass -> AO kernel: one record per shaded hit (NULL: AO rays traced inline)
  uint32_t          *aoCount;       // surfaces pre-pass -> AO kernel: number of listed hits (cleared before the pre-pass)
  // option ao_defer = 2: the listed rays sorted by (pixel block of the hit | direction class) before they are traced
  uint32_t          *aoKeys;        // per ray (2 per hit): its bin (NULL: rays traced in list order)
  uint32_t          *aoHist;        // per bin: count, then (scanned) the bin's cursor into aoOrder
  uint32_t          *aoOrder;       // ray indices in bin order
  uint8_t           *aoHit;         // per ray: hit flag
  uint32_t           aoBins;
  uint32_t          *tileCostPre;   // != null: per tile id, steps of the longest iso march of the surfaces pre-pass
  uint32_t          *tileCost;      // != null: per tile id, brick visits of the tile's longest ray (launch-order feedback)
};

// ---- exa_lbvh.hip: LBVH topology over numPrims boxes (6 floats each, device), built on the device ----
hipError_t buildLbvhTopologyDevice(const float *boxes, uint32_t numPrims, BvhNode *nodes, int32_t *levelIds,
                                   std::vector<uint32_t> &internalNodesPerDepth, hipStream_t s);

// ---- launchers implemented in exa_kernels.hip ----
// The kernels that evaluate the hat basis are compiled twice, once per association of the eight-corner sums of
// addBasisFunctions (exabrick.cu:620-777): namespace form0 = the reference's source order (-DEXA_BASIS_FORM=0, the
// definition), form1 = per axis with fused multiply-adds (-DEXA_BASIS_FORM=1; option "basis_form", oracle:
// or_set_basis_form).  One translation unit each (exa_kernels_f0.o / exa_kernels_f1.o).
#define EXA_FORM_LAUNCHERS                                                                                              \
  hipError_t launchRender(const RenderArgs &a, int numBlocks, bool grad, bool iso, bool stats, hipStream_t s);          \
  hipError_t launchSurfacePrepassKd(const RenderArgs &a, int numBlocks, bool stats, hipStream_t s);                     \
  /* the deferred AO rays of that pre-pass' hits; the finish of the pixels of a march that stored its colour (pixOut) */ \
  hipError_t launchAoRaysKd(const RenderArgs &a, int numBlocks, hipStream_t s);                                         \
  hipError_t launchCompositeKd(const RenderArgs &a, int numBlocks, hipStream_t s);                                      \
  hipError_t launchRenderKd(const RenderArgs &a, int numBlocks, bool grad, bool fast, bool surf,                        \
                            int stats /*0, 1 counters, 2 phase times*/, hipStream_t s);                                 \
  /* the same march on the rope walk (a.ropeLeaves) */                                                                  \
  hipError_t launchRenderKdRope(const RenderArgs &a, int numBlocks, bool grad, bool fast, bool surf,                    \
                                int stats, hipStream_t s);                                                              \
  hipError_t launchRenderKdWide(const RenderArgs &a, int numTiles, int lanesPerRay, bool grad, bool fast, bool surf,    \
                                hipStream_t s);                                                                         \
  /* computeTraces (exabrick.cu:1531-1574): one thread per trace, run before the frame kernel */                        \
  hipError_t launchComputeTraces(const RenderArgs &a, float *traces, int count, hipStream_t s);
namespace form0 { EXA_FORM_LAUNCHERS }
namespace form1 { EXA_FORM_LAUNCHERS }
// ... and once more in the source order with the reference's ALLOW_EMPTY_CELLS semantics (-DEXA_EMPTY_CELLS=1: a corner whose
// cell holds EXA_EMPTY_CELL_POISON_VALUE is skipped, exabrick.cu:614-618), for scenes marked allowEmptyCells
// (exa_kernels_f0e.o; no interleaved and no wide march there)
namespace form0e { EXA_FORM_LAUNCHERS }
#undef EXA_FORM_LAUNCHERS

// ---- kernels that never sample (compiled once, with form 0) ----
hipError_t launchVolumeActivity(const DeviceScene &sc, const ExaHipFrameState &fs, const ExaHipParams &p,
                                const float4 *xf, uint8_t *active, float tfFracMagic, hipStream_t s);
hipError_t launchIsoActivity(const DeviceScene &sc, const ExaHipFrameState &fs, uint8_t *active, hipStream_t s);
// refit one height class of internal nodes: box of each child = union below it
hipError_t launchRefit(BvhNode *nodes, const int32_t *nodeIds, int count, const float *domain,
                       const uint8_t *active, hipStream_t s);
// kd activity bits of one height class; which = 0 volume, 1 iso
hipError_t launchKdRefit(KdNodeDev *nodes, KdNodeDev *marchNodes, const int32_t *nodeIds, int count, const uint8_t *active, int which, hipStream_t s);
// activity bit `which` of the rope leaves from the per-region flags; *activeCount += number of active regions (may be NULL)
hipError_t launchRopeActivity(RopeLeaf *leaves, uint32_t numRegions, const uint8_t *active, int which, uint32_t *activeCount, hipStream_t s);
// brick b's cells from srcBegin[b] to dstBegin[b] in every field (src -> dst), then `begin` of every brick record / march header
hipError_t launchPermuteBricks(const float *src, float *dst, const uint32_t *srcBegin, const uint32_t *dstBegin, int4 *bricks,
                               unsigned long long numBricks, int4 *leafHdr, const int32_t *leafList, unsigned long long leafListSize,
                               unsigned long long totalCells, int numFields, hipStream_t s);
// out[cell * nch + c] = scalars[channelOffset[c] + cell] for c < nch
hipError_t launchInterleave(const DeviceScene &sc, unsigned long long totalCells, int nch, float *out, hipStream_t s);
hipError_t launchUntile(const uint32_t *gathered, unsigned long long shardStride, int world,
                        int W, int H, uint32_t *out, hipStream_t s);
hipError_t launchProfileMarker(int tag, hipStream_t s);

} // namespace exa
