/* exa_hip.h — C ABI of the MI355X-native ExaBrick render module (libexa_hip.so).
 *
 * This is the drop-in boundary for the reference's `exa::OptixRenderer`
 * (exa/OptixRenderer.h:32-97) and the device programs it launches
 * (programs/exabrick.cu).  Plain pointers and sizes only; no C++/torch types.
 * The C++ facade `exa::Renderer` (owlexabrick_amd/host/exa_host.h) keeps the
 * OptixRenderer method names on top of these entry points; INTEGRATION.md shows
 * the binding a maintainer of the reference would add.
 *
 * Two groups:
 *   exa_prep_*  host-side data preparation the OptixRenderer constructor does
 *               (brick flattening, scalar gather, same-bricks regions)
 *   exa_hip_*   the device module (upload, LBVH, activity, render, readback)
 *
 * All functions return 0 on success, non-zero on error; the message is
 * available from exa_hip_last_error()/exa_prep_last_error().  Nothing here
 * falls back to a CPU renderer: without a HIP device exa_hip_create fails.
 */
#ifndef EXA_HIP_H
#define EXA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EXA_NUM_XF_VALUES      128 /* exa/common.h:41 */
#define EXA_MAX_CHANNELS       10  /* exa/common.h:42 */
#define EXA_MAX_ISO_SURFACES   2   /* exa/common.h:43 */
#define EXA_MAX_CONTOUR_PLANES 3   /* exa/common.h:44 */

/* ---- POD mirrors of the reference's shared host/device structs ---- */

/* programs/Brick.h:31-71 */
typedef struct ExaBrick {
  int32_t  lower[3];
  int32_t  size[3];
  int32_t  level;
  uint32_t begin;   /* offset of the brick's first cell in the gathered scalar arrays */
} ExaBrick;

/* exa/Regions.h:31-41 (ExaBrickRegions::BrickRegion == SameBricksRegion) */
typedef struct ExaBrickRegion {
  float   domain_lo[3], domain_hi[3];
  float   valueRange_lo, valueRange_hi;
  int32_t leafListBegin;
  int32_t leafListSize;
  float   finestLevelCellWidth;
} ExaBrickRegion;

/* programs/FrameState.h:29-71.  The per-channel cudaTextureObject_t handles are
 * replaced by exa_hip_set_xf(); bools are int32. */
typedef struct ExaHipFrameState {
  float cam_pos[3], cam_dir00[3], cam_dirDu[3], cam_dirDv[3];
  struct { int32_t enabled; float value; int32_t channel; } iso[EXA_MAX_ISO_SURFACES];
  struct { int32_t enabled; float normal[3]; int32_t channel; float offset; } contour[EXA_MAX_CONTOUR_PLANES];
  struct { float lo[3], hi[3]; int32_t enabled; } clipBox;
  struct { float length; int32_t enabled; } ao;
  float   clockScale;
  float   xfm_vx[3], xfm_vy[3], xfm_vz[3], xfm_p[3]; /* affine3f voxelSpaceTransform */
  int32_t frameID;
  float   xfDomain[EXA_MAX_CHANNELS][2];
  float   xfOpacityScale;
} ExaHipFrameState;

/* the scalar launch parameters of programs/LaunchParams.h:26-80 the path reads,
 * plus VolumeData.numChannels / spaceSkippingEnabled (programs/VolumeData.h:24-31) */
typedef struct ExaHipParams {
  float   dt;                   /* OptixRenderer::updateDt            (OptixRenderer.cpp:413-416) */
  int32_t numPrimaryChannels;   /* multiFieldDvr ? #fields : 1        (OptixRenderer.cpp:284)     */
  int32_t colormapChannel;      /*                                    (OptixRenderer.cpp:278-283) */
  int32_t gradientShadingDVR;   /* setGradientShadingDVR              (OptixRenderer.cpp:434-437) */
  int32_t gradientShadingISO;   /* setGradientShadingISO              (OptixRenderer.cpp:439-442) */
  int32_t numChannels;          /* VolumeData.numChannels             (OptixRenderer.cpp:650)     */
  int32_t spaceSkippingEnabled; /* !contourPlanesActive && doSpaceSkipping (OptixRenderer.cpp:418-432) */
} ExaHipParams;

/* The recursion of ExaBrickRegions::buildRec (exa/Regions.cpp:73-179) is a kd-tree
 * whose leaves are the regions; exa_prep keeps it.  A child reference is >= 0 for a
 * node index, < 0 for a leaf (region id = ~ref), EXA_KD_EMPTY for an empty side.
 * Optional input of exa_hip_create: with it the module walks the regions in exact
 * front-to-back order; without it (regions built elsewhere) it uses its LBVH. */
#define EXA_KD_EMPTY INT32_MIN
typedef struct ExaKdNode {
  float   split;   /* plane position on `axis`                         */
  int32_t axis;    /* 0,1,2                                            */
  int32_t left;    /* child on the lower side of the plane             */
  int32_t right;   /* child on the upper side                          */
} ExaKdNode;

/* what the OptixRenderer constructor uploads (OptixRenderer.cpp:95-98,133-141,160-168) */
typedef struct ExaHipScene {
  const ExaBrick       *bricks;        uint64_t numBricks;
  const ExaBrickRegion *regions;       uint64_t numRegions;
  const int32_t        *leafList;      uint64_t leafListSize;
  const float          *scalars;       /* numFields * totalCells floats, brick order    */
  const uint64_t       *channelOffset; /* numFields entries (reference: 32-bit unsigned) */
  uint64_t              totalCells;
  int32_t               numFields;
  float                 voxelBounds_lo[3], voxelBounds_hi[3];
  const ExaKdNode      *kdNodes;       /* optional (may be NULL) */
  uint64_t              numKdNodes;
  int32_t               kdRoot;        /* reference of the root (a leaf ref for a one-region scene) */
  /* (added in round 4, at the end: a caller that fills the struct by hand zero-initialises it first; exa_hip_create refuses
     any value other than 0 and 1, so that a struct filled field by field without this one is caught, not misread) */
  int32_t               allowEmptyCells; /* the reference's compile-time option ALLOW_EMPTY_CELLS (CMakeLists.txt:70-73, default
                                          OFF) as a property of the scene: scalars equal to EXA_EMPTY_CELL_POISON_VALUE are
                                          "no cell here" and addBasisFunctions skips them (programs/exabrick.cu:614-618) */
} ExaHipScene;

/* programs/FrameState.h:27 */
#define EXA_EMPTY_CELL_POISON_VALUE (-1e20f)
/* exa_prep_create_ex flags */
#define EXA_PREP_ALLOW_EMPTY_CELLS 1   /* a negative cell id = no cell: its slot holds the poison value (exa/OptixRenderer.cpp:116-118) */

/* work counters of one frame (instrumented kernel variant); the basis of the
 * algorithmic-bytes figure in DESIGN.md */
typedef struct ExaHipStats {
  uint64_t segments;      /* traceVolumeRay hits                                  */
  uint64_t sample_evals;  /* samplePoint[WithDerivative] calls in the DVR march    */
  uint64_t samples;       /* ...of which valid                                     */
  uint64_t brick_visits;  /* addBasisFunctions calls, all paths                    */
  uint64_t corner_loads;  /* cell scalars read, all paths                          */
  uint64_t iso_segments;  /* iso-BVH hits                                          */
  uint64_t iso_evals;     /* sample calls of the iso march (+ re-samples)          */
  uint64_t nodes_visited; /* acceleration-structure nodes fetched: 64-B LBVH nodes or
                             16-B kd nodes, see node_bytes                         */
  uint64_t node_bytes;    /* bytes per node of the structure that was walked       */
  uint64_t pixels;        /* pixels rendered by this handle (its tile shard)       */
  uint64_t diag[9];       /* kd kernel diagnostics: {waves,lanes} x {brick visit, sample epilogue,
                             kd node step, leaf accept}, then kd-interval/slab-test mismatches */
  uint64_t phase_cycles[5]; /* kd kernel: shader-clock cycles of its waves, summed, by phase: brick visit, sample
                             epilogue, kd walk, segment pop, other (ray set-up, output) */
  float    kernel_ms;     /* hipEvent time of the last render launch               */
  float    rebuild_ms;    /* hipEvent time of the last activity+refit pass         */
  uint64_t walk_restarts;     /* kd walk: restarts from the root after the 4-entry short stack dropped an entry */
  uint64_t walk_union_nodes;  /* option walk_probe: kd nodes visited, counted once per WAVE (the union over its 64 rays):
                                 what a wave-coherent (packet) walk would have to step through at least */
  uint64_t walk_probe_overflow; /* ... lanes that found their wave's probe table full (0 in a valid measurement) */
  uint64_t wave_iters;        /* kd march: march iterations of every wave's longest ray, summed over the waves ...       */
  uint64_t tile_iters;        /* ... and 4 x the slowest wave's per workgroup, summed: wave_iters / tile_iters = how evenly
                                 the four waves of a workgroup finish (its LDS is held until the slowest one does)        */
  uint64_t walk_leaf_visits;  /* rope walk (option "walk"): leaves fetched, 64 B each — counted as four 16-byte nodes in
                                 nodes_visited, so nodes_visited - 4 * walk_leaf_visits inner nodes were stepped through;
                                 0 for the stack walk, whose leaves are references inside their parent node              */
} ExaHipStats;

typedef struct ExaHipRenderer ExaHipRenderer;
typedef struct ExaPrep ExaPrep;

/* ------------------------------------------------------------------ */
/* host data preparation                                               */
/* ------------------------------------------------------------------ */

/* OptixRenderer::OptixRenderer data prep (exa/OptixRenderer.cpp:71-141):
 * brick flattening with running `begin`, index-vector concat, per-field gather
 * scalar[i] = field[cellID[i]]; then ExaBrickRegions::buildFrom
 * (exa/Regions.cpp:242-320) over the first numRegionFields fields.
 * bricks7 = numBricks x {size.xyz, lower.xyz, level}, the `.bricks` record
 * header order (exa/ExaBricks.cpp:27-33).  Errors mirror the reference's
 * std::runtime_error texts. */
int exa_prep_create(const int32_t *bricks7, uint64_t numBricks,
                    const int32_t *cellIDs, uint64_t numCellIDs,
                    const float *const *fields, const uint64_t *fieldLen,
                    int32_t numFields, int32_t numRegionFields, int32_t numThreads,
                    ExaPrep **out);
/* as exa_prep_create with `flags` (EXA_PREP_*).  EXA_PREP_ALLOW_EMPTY_CELLS = the reference built with
 * -DALLOW_EMPTY_CELLS=1: every negative cell id is "no cell", as in the renderer (exa/OptixRenderer.cpp:116-118; that only -1
 * occurs is an assert of the loader, exa/ExaBricks.cpp:46-49, compiled out of a release build); the regions' value ranges include the poison
 * value, as the reference's computeValueRange (exa/Regions.cpp:182-240) does; the scene is marked allowEmptyCells */
int exa_prep_create_ex(const int32_t *bricks7, uint64_t numBricks,
                       const int32_t *cellIDs, uint64_t numCellIDs,
                       const float *const *fields, const uint64_t *fieldLen,
                       int32_t numFields, int32_t numRegionFields, int32_t numThreads, int32_t flags,
                       ExaPrep **out);
void exa_prep_destroy(ExaPrep *);
/* fills an ExaHipScene whose pointers stay valid until exa_prep_destroy */
int  exa_prep_scene(const ExaPrep *, ExaHipScene *out);
const char *exa_prep_last_error(void);
/* diagnostic: the leaves and neighbour links exa_hip builds for its rope walk (option "walk") from this scene's kd-tree, on the
 * host (owlexabrick_amd/csrc/exa_ropes.h).  Call with leafBoxes == NULL for the counts, then with arrays of that size:
 * leafBoxes numLeaves x 6 floats (lo, hi), leafLinks numLeaves x 6 (-x +x -y +y -z +z: inner node >= 0, leaf ~index, outside
 * the root box = EXA_KD_EMPTY + 1), leafRegion numLeaves (region id; -1: a gap = an empty child slot of the tree), nodes
 * numNodes (the tree behind the links: no empty slots).  *flags: bit 0 = every region leaf's box is its domain, bit 1 = the
 * planes are in the range of the walk's short exact division. */
int exa_prep_ropes(const ExaPrep *, uint64_t *numLeaves, uint64_t *numNodes, float *leafBoxes, int32_t *leafLinks,
                   int32_t *leafRegion, ExaKdNode *nodes, int32_t *flags);

/* ------------------------------------------------------------------ */
/* device module                                                       */
/* ------------------------------------------------------------------ */

/* replaces the upload + accel half of OptixRenderer::OptixRenderer
 * (exa/OptixRenderer.cpp:95-98,133-141,160-168,305-316): copies the scene to HBM
 * on `device`, builds the LBVH over the regions.  The caller's arrays are not
 * referenced after return. */
int exa_hip_create(const ExaHipScene *scene, int32_t device, ExaHipRenderer **out);
/* One handle that drives several GPUs of one node (SURVEY 8(b) threading row, 8(e)); new relative to the reference,
 * whose raygen program never reads LaunchParams.deviceIndex/deviceCount (programs/LaunchParams.h:31-32).  The scene is
 * replicated on every device of the list; device i renders the 16x16 tiles t with t % numDevices == i on a stream of
 * its own and stores them straight into the destination frame, which lives on devices[0] (the other devices write it
 * through a peer mapping over xGMI): no gather, no untile.  Every other entry point takes the handle unchanged
 * (exa_hip_set_shard is refused).  A device destination passed to exa_hip_render must be memory of devices[0]; with
 * async != 0 the call returns once the work is queued and `hipStream` (a stream of devices[0]) waits for all
 * devices, so the caller can queue the frame's copy-out behind it and start the next frame into another buffer.
 * Entries of `devices` may repeat (several renderers on one GPU: rehearsal on a one-GPU box). */
int exa_hip_create_multi(const ExaHipScene *scene, const int32_t *devices, int32_t numDevices, ExaHipRenderer **out);
int exa_hip_destroy(ExaHipRenderer *);

/* OptixRenderer::resizeFrameBuffer (exa/OptixRenderer.cpp:341-355): (re)allocates
 * the float4 accumulation buffer; the colour destination is passed per render. */
int exa_hip_resize(ExaHipRenderer *, int32_t width, int32_t height);

/* whole-struct upload as every OptixRenderer setter does (updateCamera,
 * updateFrameID, updateIsoValues, setVoxelSpaceTransform ...;
 * exa/OptixRenderer.cpp:322-339,357-368,406-411,489-502).  A change of
 * xfDomain/xfOpacityScale marks the volume LBVH dirty, a change of iso[] marks
 * the iso LBVH dirty (needVolumeBVHRebuild / needIsoBVHRebuild). */
int exa_hip_set_frame_state(ExaHipRenderer *, const ExaHipFrameState *);

/* OptixRenderer::updateXF texture upload (exa/OptixRenderer.cpp:385-402):
 * 128 x (r,g,b,a) for channel `chan`; marks the volume LBVH dirty. */
int exa_hip_set_xf(ExaHipRenderer *, int32_t chan, const float *rgba128);

/* the triangle surfaces of the OptixRenderer constructor (createSurfaces, exa/OptixRenderer.cpp:554-612;
 * closest-hit shading programs/exabrick.cu:420-433): all meshes concatenated, world space,
 * vertices 3 floats each, triangles 3 vertex indices each.  numTris = 0 removes them. */
int exa_hip_set_triangles(ExaHipRenderer *, const float *vertices, uint64_t numVertices,
                          const int32_t *triangles, uint64_t numTris);

/* streamline tracer (OptixRenderer::traces, exa/OptixRenderer.h:160-170) */
typedef struct ExaHipTracer {
  int32_t enabled;          /* setTracerEnabled                                   */
  int32_t channels[3];      /* the three scalar fields read as a velocity          */
  int32_t numTraces, numTimesteps;
  float   steplen;
} ExaHipTracer;
/* resetTracer (exa/OptixRenderer.cpp:450-472): seeds (numTraces x 3, the caller draws them) become
 * timestep 0 of every trace, the rest is cleared, the current timestep returns to 0 */
int exa_hip_reset_tracer(ExaHipRenderer *, const ExaHipTracer *, const float *seeds);
int exa_hip_set_tracer_enabled(ExaHipRenderer *, int32_t enabled);
/* advanceTracer (:474-487): next timestep; *rebuild = needStreamlineBVHRebuild */
int exa_hip_advance_tracer(ExaHipRenderer *, int32_t *rebuild);
/* numTraces * numTimesteps * 3 floats */
int exa_hip_read_traces(ExaHipRenderer *, float *dst);

/* updateDt / setSpaceSkipping / setGradientShading* (exa/OptixRenderer.cpp:413-442) */
int exa_hip_set_params(ExaHipRenderer *, const ExaHipParams *);

/* image-space sharding for multi-GPU: this handle renders the 16x16-pixel tiles
 * t with t % worldSize == rank (row-major tile order) and writes them compactly,
 * tile-major, 256 pixels per tile.  rank 0 / worldSize 1 = whole frame in the
 * normal row-major layout.  New relative to the reference (it never shards). */
int exa_hip_set_shard(ExaHipRenderer *, int32_t rank, int32_t worldSize);
/* number of uint32 pixels exa_hip_render writes for the current size/shard */
uint64_t exa_hip_output_pixels(const ExaHipRenderer *);
/* root side of the gather: `gathered` holds worldSize shards back to back, each
 * padded to shardStridePixels; writes the row-major W*H image.  Device pointers. */
int exa_hip_untile(ExaHipRenderer *, const uint32_t *gathered, uint64_t shardStridePixels,
                   int32_t worldSize, uint32_t *rgba8_out, void *hipStream);

/* OptixRenderer::render (exa/OptixRenderer.cpp:531-552): re-evaluates region
 * activity and refits the dirty LBVH(s), then launches the frame.
 * rgba8 is the caller-owned colour buffer (resizeFrameBuffer's fbPointer):
 * a device pointer if dstIsDevice, else host memory (copied back synchronously).
 * hipStream: a hipStream_t or NULL for the default stream.  Synchronous unless
 * dstIsDevice and async != 0. */
int exa_hip_render(ExaHipRenderer *, uint32_t *rgba8, int32_t dstIsDevice,
                   void *hipStream, int32_t async);

/* same frame through the instrumented kernel variant; fills counters */
int exa_hip_render_stats(ExaHipRenderer *, uint32_t *rgba8, int32_t dstIsDevice, ExaHipStats *out);
int exa_hip_get_stats(ExaHipRenderer *, ExaHipStats *out);

/* accumulation buffer (float4 per pixel, row-major, this shard's layout) */
int exa_hip_read_accum(ExaHipRenderer *, float *dst4);
int exa_hip_write_accum(ExaHipRenderer *, const float *src4);

/* region activity as the VolumeBVH / IsoSurface bounds programs see it
 * (programs/exabrick.cu:285-312, 373-402), one byte per region; for tests */
int exa_hip_read_activity(ExaHipRenderer *, int32_t which /*0 volume, 1 iso*/, uint8_t *dst);

/* tuning knobs that never change results: "tile_order" = launch sequence of the 16x16 tiles:
 * 0 row-major, 1 row-major 8x8 supertiles per XCD, 2 pseudo-random, 3 centre-out, 4 Z-order
 * (default), 5/6/7 Z-order dealt to the XCDs in chunks of 16/64/256 tiles; "accel" 0 = LBVH with restart per segment,
 * 1 = region kd-tree walked front to back (default when the scene carries one); "lbvh_build" 0 (default) = that LBVH
 * is built on the device (Morton codes, radix sort, topology level by level), 1 = the same tree built on the host;
 * "fast_sampler" 1 (default) = the surfaces pre-pass of the kd path samples through the march headers with the
 * masked-weight basis evaluation, 0 = with the literal addBasisFunctions (same sums bit for bit); "interleave" 1 (default) =
 * a DVR march of 2..4 primary channels reads a channel-interleaved copy float[cell][channel] of those fields (built on
 * the device at the first such frame) and evaluates all channels per brick visit, 0 = field by field from the arrays
 * as uploaded; "addr64" 1 = the march forms 64-bit cell / header / node addresses even where a scene is small enough for
 * 32-bit offsets from a uniform base (default 0: chosen per scene; "pack_records" 0 = the march takes region ids from the walk and loads the region
 * records, as it does in scenes whose records {first brick, brick count, level} do not fit the 32 bits of a leaf reference (default 1:
 * packed where they fit; tests); tests); "brick_order" 0 = the bricks' cells lie in
 * memory in the order of the brick list (the running `begin` of OptixRenderer.cpp:71-93), 1 = along a Morton curve of the
 * brick centres (re-laid on the device at the next frame; cells are only found through their brick's `begin`, so pixels
 * cannot change; environment EXA_BRICK_ORDER sets the initial value); "tile_feedback" 1 (default) = after a
 * change of view / TF / layout the next synchronous frame records every tile's longest ray and later frames launch
 * the heaviest tiles first (a frame's critical path is its longest rays), 0 = keep the static order; "wide_march" 1
 * (default) = tiles whose longest ray would outlast the rest of the frame (multi-GPU shards) march with 2 or 4 lanes
 * per ray — the walk split into depth windows, consecutive samples evaluated side by side and composited in order,
 * bit-identical pixels; up to 8 GiB of device memory for the walkers' leaf lists — 0 = never,
 * 2 / 4 = every tile with that many lanes (tests); "prepass_split" 1 (default) = in a frame with surfaces the tiles whose iso marches are
 * long (measured by the same frame that measures the tile costs) get their own pre-pass + march pipeline on a side stream,
 * beside the pre-pass + march of the other tiles (the pre-pass is bound by the latency of its longest rays, the march by
 * throughput), 0 = the whole pre-pass in front of the whole march; "ao_defer" 1 (default) = the ambient-occlusion rays of the shaded hits are traced by a launch of their own,
 * one ray per lane over a compact list of the hits, 2 = as 1 with the listed rays sorted on the device by (32x32-pixel block of the
 * hit | direction class: octant x dominant axis) before they are traced, so that a wave's 64 rays start close together and head
 * the same way (counting sort: histogram, scan, scatter; hit flags combined per hit by a last kernel), 0 = inline behind each pixel's primary ray (1 is the default); "ao_overlap" 1 = the deferred AO rays run on a side
 * stream BESIDE the march instead of in front of it: the march needs the surfaces' hit distance up front but their colour only for
 * its last operation, so it stores its pixel colour and a small kernel finishes the pixels (over the surfaces' colour, accumulation,
 * sRGB, pack: the same operations in the same order) once both are done, 0 (default: the faster one with several frames in flight) = pre-pass, AO rays, march one after the other; "stats_mode" = what exa_hip_render_stats collects: 1 (default) the
 * work counters, 2 only phase_cycles, from the shipped code plus a clock read at every phase change; "walk_probe" 1 = the
 * counting variant also records every wave's SET of visited kd nodes (128 KiB of device memory per wave) and reports its
 * size summed over the waves as walk_union_nodes (a diagnostic of how coherent the 64 walks of a wave are);
 * "profile_marker" N = launch an empty kernel (profileMarkerKernel) on the null stream now: a bracket in a profiler's
 * dispatch list, no effect on any frame.
 * "walk" selects how the DVR march of the kd path finds its segments: 1 = the ordered walk of the region kd-tree with a
 * 4-entry short stack in LDS (restart from the root when an entry was dropped), which skips subtrees without an active
 * region; 2 = the rope walk: every leaf carries its box and one link per face to its neighbour (64 B per leaf, built on
 * the host at the first frame that uses it), the ray goes from leaf to leaf, the reference's slab test is evaluated on
 * each leaf's own box, no stack and no restarts — and the LDS of the stack goes to a segment queue of five 16-byte entries
 * (region record, t1, the first sample's t_i, t0) instead of four 12-byte ones; inactive leaves are passed through one by one; 0 (default) = per frame: the rope walk when at least 40 % of the
 * regions are active for the volume march, else the stack walk.  Same segments, same pixels either way.
 * "basis_form" selects the association of the eight-corner sums of addBasisFunctions (programs/exabrick.cu:620-777):
 * 1 (default) = per axis — x-pairs, then y, then z, weight sums as products of per-axis sums — with fused multiply-adds
 * (49 instead of 116 floating-point operations per brick with derivatives), 0 = the reference's source order with every
 * product and sum rounded separately.  The reference binary computes neither literally (nvcc contracts a*b+c by
 * default and CMakeLists.txt passes no -fmad=false); the CPU oracle restates both operation for operation
 * (or_set_basis_form) and the kernels equal it in either.  Form 1 against form 0: |d accum| <= 1e-3, RGBA8 <= 1 LSB
 * (tests/test_basis_form.py).  Environment EXA_BASIS_FORM sets the initial value.
 * Two knobs move results within the stated float tolerance: "fast_math" 1 (default)
 * evaluates the opacity correction powf as exp2(dt*log2(x)) on the hardware
 * transcendental units (~2 ulp), 0 uses the library powf (<1 ulp); "tf_filter" 1 (default) holds the
 * transfer-function filter weight in 9-bit fixed point with 8 fractional bits, as the CUDA texture unit
 * behind the reference's tex1D<float4> fetch does (programs/exabrick.cu:147, exa/Texture.h:141-147; CUDA C
 * programming guide, "Texture Fetching", linear filtering), 0 keeps the full-precision weight. */
int exa_hip_set_option(ExaHipRenderer *, const char *key, int32_t value);

const char *exa_hip_last_error(const ExaHipRenderer * /* may be NULL: creation errors */);

#ifdef __cplusplus
}
#endif
#endif
