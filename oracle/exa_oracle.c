/* exa_oracle.c — CPU ORACLE (test infrastructure only; see exa_oracle.h).
 *
 * Plain-C restatement of the reference hot path.  Citations are relative to the
 * reference tree.  Build with -ffp-contract=off so every a*b+c below is two
 * IEEE binary32 operations, exactly as written.
 *
 * PARITY STATUS: parity unpinned vs the real OptiX renderer (no goldens exist
 * upstream); pinned by analytic KATs, committed oracle-generated fixtures, and
 * four independent numpy restatements of the reference's text (one-region DVR
 * pixel incl. gradient shading, the region loop on the hat-basis definition, the
 * implicit iso-surface, the hat-basis sample) swept over thousands of seeded
 * random scenes (tests/fuzz_oracle.py, tests/fuzz_spec*.py).
 */
#include "exa_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* vec3f helpers — restating the un-vendored owl::common math          */
/* (componentwise ops, left-to-right dot, embree-style madd xfm).      */
/* ------------------------------------------------------------------ */
typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3s(float s) { return V3(s, s, s); }
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vdiv(v3 a, v3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 vscale(float s, v3 a) { return V3(s * a.x, s * a.y, s * a.z); }
static inline v3 vneg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float vlength(v3 a) { return sqrtf(vdot(a, a)); }
/* the same sum, left to right, with both additions fused into their products (basis form 1) */
static inline float vdotf(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
/* owl normalize(v) = v * rsqrt(dot(v,v)); host rsqrt(f) = 1.f/sqrtf(f) */
static inline v3 vnormalize(v3 a) { return vscale(1.f / sqrtf(vdot(a, a)), a); }
static inline v3 vcross(v3 a, v3 b)
{ return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline v3 vfrom(const float *p) { return V3(p[0], p[1], p[2]); }
static inline float vget(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

/* owl xfmPoint(m,p) = madd(p.x,vx, madd(p.y,vy, madd(p.z,vz, m.p))) */
static inline v3 xfm_point(const OrFrameState *fs, v3 p)
{
  v3 vx = vfrom(fs->xfm_vx), vy = vfrom(fs->xfm_vy), vz = vfrom(fs->xfm_vz), P = vfrom(fs->xfm_p);
  return vadd(vscale(p.x, vx), vadd(vscale(p.y, vy), vadd(vscale(p.z, vz), P)));
}
/* owl xfmVector(m,v) = madd(v.x,vx, madd(v.y,vy, v.z*vz)) */
static inline v3 xfm_vector(const OrFrameState *fs, v3 v)
{
  v3 vx = vfrom(fs->xfm_vx), vy = vfrom(fs->xfm_vy), vz = vfrom(fs->xfm_vz);
  return vadd(vscale(v.x, vx), vadd(vscale(v.y, vy), vscale(v.z, vz)));
}

/* ------------------------------------------------------------------ */
/* owl::common::LCG<16> (owl/common/math/random.h, un-vendored):       */
/* TEA-style 16-round seed mix, then state = 1664525*state+1013904223, */
/* returned as (state & 0x00FFFFFF) / 2^24.                            */
/* ------------------------------------------------------------------ */
typedef struct { uint32_t state; } Lcg;
static inline void lcg_init(Lcg *r, uint32_t val0, uint32_t val1)
{
  uint32_t v0 = val0, v1 = val1, s0 = 0;
  for (unsigned n = 0; n < 16; n++) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
  }
  r->state = v0;
}
static inline float lcg_next(Lcg *r)
{
  r->state = 1664525u * r->state + 1013904223u;
  return (float)(r->state & 0x00FFFFFFu) / (float)0x01000000;
}

float or_lcg_init_next(uint32_t seed0, uint32_t seed1, int ndraws, float *draws)
{
  Lcg r; lcg_init(&r, seed0, seed1);
  float last = 0.f;
  for (int i = 0; i < ndraws; i++) { last = lcg_next(&r); if (draws) draws[i] = last; }
  return last;
}

/* ------------------------------------------------------------------ */
/* scene                                                               */
/* ------------------------------------------------------------------ */
typedef struct {           /* one call of buildRec that produced children or a leaf */
  float lo[3], hi[3];      /* the 'domain' argument                                  */
  int32_t left, right;     /* child node ids (-1 none)                               */
  int32_t region;          /* >=0 for a leaf that produced a region                  */
} KdNode;

struct OrScene {
  size_t   numBricks, totalCells;
  OrBrick *bricks;
  int      numFields;
  float   *scalars;        /* numFields * totalCells                   */
  size_t  *offsets;        /* f * totalCells (reference: unsigned)     */
  OrRegion *regions; size_t numRegions, capRegions;
  int32_t  *leafList; size_t numLeaf, capLeaf;
  KdNode   *nodes; size_t numNodes, capNodes;
  float     xf[OR_MAX_CHANNELS][OR_NUM_XF_VALUES][4];
  float     tfFracMagic;   /* 32768: filter fraction held in 1.8 fixed point (CUDA tex1D); 0: full precision */
  int       allowEmptyCells; /* the reference's compile-time option ALLOW_EMPTY_CELLS (CMakeLists.txt:70, default OFF): cell id -1 =
                                no cell; its slot holds EMPTY_CELL_POISON_VALUE and addBasisFunctions skips it */
  int       basisForm;     /* 0: addBasisFunctions in the reference's source order (the definition, default);
                              1: the same eight-corner sums associated per axis, x -> y -> z, with fused multiply-adds */
  float    *meshVerts; int32_t *meshTris; size_t numVerts, numTris;   /* all surfaces, concatenated */
  /* streamline tracer state (OptixRenderer.h:160-170) */
  int tracerEnabled, tracerChannels[3], numTraces, numTimesteps;
  int timestepHost;   /* traces.timestepHost: counts every advanceTracer call */
  int timestep;       /* traces.currentTimestep, what the device programs read: uploaded only while <= numTimesteps */
  float steplen; float *traces;                                        /* numTraces*numTimesteps*3 */
  float     vb_lo[3], vb_hi[3];
};

typedef struct { float lo[3], hi[3]; int32_t id; } Prim;

static void die_oom(void) { fprintf(stderr, "exa_oracle: out of memory\n"); abort(); }
static void *xmalloc(size_t n) { void *p = malloc(n ? n : 1); if (!p) die_oom(); return p; }
static void *xrealloc(void *q, size_t n) { void *p = realloc(q, n ? n : 1); if (!p) die_oom(); return p; }

static int cmp_i32(const void *a, const void *b)
{ int32_t x = *(const int32_t *)a, y = *(const int32_t *)b; return (x > y) - (x < y); }

/* exa/Regions.cpp:32-71 addLeaf */
static int32_t add_leaf(OrScene *S, const Prim *prims, size_t n, const float lo[3], const float hi[3])
{
  if (lo[0] >= hi[0]) return -1;
  if (lo[1] >= hi[1]) return -1;
  if (lo[2] >= hi[2]) return -1;
  /* std::set<int> allBrickIDs -> ascending distinct ids (:44-46) */
  int32_t *ids = (int32_t *)xmalloc(n * sizeof(int32_t));
  for (size_t i = 0; i < n; i++) ids[i] = prims[i].id;
  qsort(ids, n, sizeof(int32_t), cmp_i32);
  size_t m = 0;
  for (size_t i = 0; i < n; i++) if (m == 0 || ids[m - 1] != ids[i]) ids[m++] = ids[i];
  if (m == 0) { free(ids); return -1; }

  if (S->numRegions == S->capRegions) {
    S->capRegions = S->capRegions ? 2 * S->capRegions : 1024;
    S->regions = (OrRegion *)xrealloc(S->regions, S->capRegions * sizeof(OrRegion));
  }
  if (S->numLeaf + m > S->capLeaf) {
    while (S->numLeaf + m > S->capLeaf) S->capLeaf = S->capLeaf ? 2 * S->capLeaf : 4096;
    S->leafList = (int32_t *)xrealloc(S->leafList, S->capLeaf * sizeof(int32_t));
  }
  OrRegion *R = &S->regions[S->numRegions];
  memset(R, 0, sizeof(*R));
  for (int k = 0; k < 3; k++) { R->dom_lo[k] = lo[k]; R->dom_hi[k] = hi[k]; }
  R->leafListSize = (int32_t)m;
  R->leafListBegin = (int32_t)S->numLeaf;          /* :64 */
  for (size_t i = 0; i < m; i++) S->leafList[S->numLeaf++] = ids[i]; /* :67-69 */
  free(ids);
  return (int32_t)(S->numRegions++);               /* :70 */
}

static int32_t new_node(OrScene *S, const float lo[3], const float hi[3])
{
  if (S->numNodes == S->capNodes) {
    S->capNodes = S->capNodes ? 2 * S->capNodes : 1024;
    S->nodes = (KdNode *)xrealloc(S->nodes, S->capNodes * sizeof(KdNode));
  }
  KdNode *N = &S->nodes[S->numNodes];
  for (int k = 0; k < 3; k++) { N->lo[k] = lo[k]; N->hi[k] = hi[k]; }
  N->left = N->right = N->region = -1;
  return (int32_t)(S->numNodes++);
}

/* exa/Regions.cpp:73-179 buildRec.  Takes ownership of prims (frees it, as the
 * reference clears buildPrims at :171).  Returns the kd node id (or -1). */
static int32_t build_rec(OrScene *S, Prim *prims, size_t n, const float dlo[3], const float dhi[3])
{
  if (n == 0) { free(prims); return -1; }                       /* :76 */
  for (int i = 0; i < 3; i++)
    if (dhi[i] == dlo[i]) { free(prims); return -1; }           /* :77-82 "EMPTY DOMAIN" */

  float tgtPos[3], bestPos[3], bestDist[3], span[3];
  for (int i = 0; i < 3; i++) {
    span[i] = dhi[i] - dlo[i];
    tgtPos[i] = 0.5f * (dlo[i] + dhi[i]);                       /* domain.center() :84 */
    bestPos[i] = dlo[i];                                        /* :85 */
    bestDist[i] = span[i];                                      /* :86 */
  }
  for (size_t p = 0; p < n; p++) {                              /* :89-107 */
    for (int dim = 0; dim < 3; dim++) {
      for (int side = 0; side < 2; side++) {
        float pos = side ? prims[p].lo[dim] : prims[p].hi[dim];
        if ((pos <= dlo[dim]) || (pos >= dhi[dim])) continue;
        float dist = fabsf(tgtPos[dim] - pos);
        if (dist >= bestDist[dim]) continue;
        bestPos[dim] = pos;
        bestDist[dim] = dist;
      }
    }
  }
  int splitDim = -1;
  float splitPos = 0.f;
  int widestDim = 0;                                            /* arg_max(span) :112 */
  for (int i = 1; i < 3; i++) if (fabsf(span[i]) > fabsf(span[widestDim])) widestDim = i;
  for (int i = 0; i < 3; i++) {                                 /* :113-123 */
    int dim = (widestDim + i) % 3;
    if (bestPos[dim] <= dlo[dim] || bestPos[dim] >= dhi[dim]) continue;
    splitDim = dim;
    splitPos = bestPos[dim];
    break;
  }
  int32_t me = new_node(S, dlo, dhi);
  if (splitDim < 0) {                                           /* :131-134 */
    int32_t r = add_leaf(S, prims, n, dlo, dhi);
    S->nodes[me].region = r;
    free(prims);
    return me;
  }
  float llo[3], lhi[3], rlo[3], rhi[3];
  for (int k = 0; k < 3; k++) { llo[k] = rlo[k] = dlo[k]; lhi[k] = rhi[k] = dhi[k]; }
  lhi[splitDim] = splitPos;                                     /* :138 */
  rlo[splitDim] = splitPos;                                     /* :139 */
  Prim *bl = (Prim *)xmalloc(n * sizeof(Prim)), *br = (Prim *)xmalloc(n * sizeof(Prim));
  size_t nl = 0, nr = 0;
  for (size_t i = 0; i < n; i++) {                              /* :142-169 */
    Prim c;
    c.id = prims[i].id;
    for (int k = 0; k < 3; k++) {                               /* intersection(prim, domain_l) */
      c.lo[k] = fmaxf(prims[i].lo[k], llo[k]);
      c.hi[k] = fminf(prims[i].hi[k], lhi[k]);
    }
    if (c.lo[0] < c.hi[0] && c.lo[1] < c.hi[1] && c.lo[2] < c.hi[2]) bl[nl++] = c;
    for (int k = 0; k < 3; k++) {
      c.lo[k] = fmaxf(prims[i].lo[k], rlo[k]);
      c.hi[k] = fminf(prims[i].hi[k], rhi[k]);
    }
    if (c.lo[0] < c.hi[0] && c.lo[1] < c.hi[1] && c.lo[2] < c.hi[2]) br[nr++] = c;
  }
  free(prims);                                                  /* :171 */
  /* serial_for(2): side 0 -> right, side 1 -> left (:173-178) */
  int32_t rn = build_rec(S, br, nr, rlo, rhi);
  int32_t ln = build_rec(S, bl, nl, llo, lhi);
  S->nodes[me].right = rn;
  S->nodes[me].left = ln;
  return me;
}

static inline void range_extend(OrRegion *R, float v)
{ if (v < R->vr_lo) R->vr_lo = v; if (v > R->vr_hi) R->vr_hi = v; } /* owl interval::extend */

/* exa/Regions.cpp:182-240 computeValueRange */
static void compute_value_range(OrScene *S, OrRegion *R, int numRegionFields)
{
  R->vr_lo = +INFINITY; R->vr_hi = -INFINITY;                   /* range1f() */
  for (int f = 0; f < numRegionFields; f++) {
    for (int i = 0; i < R->leafListSize; i++) {
      int brickID = S->leafList[R->leafListBegin + i];
      const OrBrick *b = &S->bricks[brickID];
      const float cellWidth = (float)(1 << b->level);
      for (int iz = 0; iz < b->size[2]; iz++) {
        float pos_z = b->lower[2] + (iz + .5f) * cellWidth;
        if (!((pos_z - cellWidth <= R->dom_hi[2]) && (pos_z + cellWidth >= R->dom_lo[2]))) continue;
        for (int iy = 0; iy < b->size[1]; iy++) {
          float pos_y = b->lower[1] + (iy + .5f) * cellWidth;
          if (!((pos_y - cellWidth <= R->dom_hi[1]) && (pos_y + cellWidth >= R->dom_lo[1]))) continue;
          for (int ix = 0; ix < b->size[0]; ix++) {
            float pos_x = b->lower[0] + (ix + .5f) * cellWidth;
            if (!((pos_x - cellWidth <= R->dom_hi[0]) && (pos_x + cellWidth >= R->dom_lo[0]))) continue;
            size_t idx = S->offsets[f] + b->begin + (size_t)ix
                       + (size_t)b->size[0] * iy + (size_t)b->size[0] * b->size[1] * iz;
            range_extend(R, S->scalars[idx]);
          }
        }
      }
    }
  }
}

OrScene *or_scene_create(const int32_t *bricks7, size_t numBricks,
                         const int32_t *cellIDs, size_t numCellIDs,
                         const float *const *fields, const size_t *fieldLen,
                         int numFields, int numRegionFields,
                         char *err, size_t errLen)
{ return or_scene_create_ex(bricks7, numBricks, cellIDs, numCellIDs, fields, fieldLen, numFields, numRegionFields, 0, err, errLen); }

/* programs/FrameState.h:27 */
#define EMPTY_CELL_POISON_VALUE -1e20f

OrScene *or_scene_create_ex(const int32_t *bricks7, size_t numBricks,
                            const int32_t *cellIDs, size_t numCellIDs,
                            const float *const *fields, const size_t *fieldLen,
                            int numFields, int numRegionFields, int allowEmptyCells,
                            char *err, size_t errLen)
{
#define FAIL(msg) do { if (err && errLen) snprintf(err, errLen, "%s", msg); or_scene_destroy(S); return NULL; } while (0)
  OrScene *S = (OrScene *)calloc(1, sizeof(OrScene));
  if (!S) die_oom();
  S->numBricks = numBricks;
  S->numFields = numFields;
  S->tfFracMagic = 32768.f;           /* default: the published CUDA linear filter */
  S->allowEmptyCells = allowEmptyCells != 0;
  S->bricks = (OrBrick *)xmalloc(numBricks * sizeof(OrBrick));
  /* exa/OptixRenderer.cpp:75-93 flatten */
  size_t scalarOffset = 0;
  S->vb_lo[0] = S->vb_lo[1] = S->vb_lo[2] = +INFINITY;
  S->vb_hi[0] = S->vb_hi[1] = S->vb_hi[2] = -INFINITY;
  for (size_t i = 0; i < numBricks; i++) {
    const int32_t *r = bricks7 + 7 * i;
    OrBrick *b = &S->bricks[i];
    b->size[0] = r[0]; b->size[1] = r[1]; b->size[2] = r[2];
    b->lower[0] = r[3]; b->lower[1] = r[4]; b->lower[2] = r[5];
    b->level = r[6];
    b->begin = (uint32_t)(int32_t)scalarOffset;
    if ((int32_t)scalarOffset < 0 || scalarOffset > 0x7fffffffull) FAIL("32-bit offset overflow"); /* :82-83 */
    size_t vol = (size_t)b->size[0] * (size_t)b->size[1] * (size_t)b->size[2];
    scalarOffset += vol;
    if (scalarOffset > numCellIDs) FAIL("failed sanity-check in brick size");   /* :89-90 */
    for (int k = 0; k < 3; k++) {                                /* ExaBricks::getBounds */
      float lo = (float)b->lower[k], hi = (float)(b->lower[k] + b->size[k] * (1 << b->level));
      if (lo < S->vb_lo[k]) S->vb_lo[k] = lo;
      if (hi > S->vb_hi[k]) S->vb_hi[k] = hi;
    }
  }
  if (scalarOffset != numCellIDs) FAIL("failed sanity-check in brick size");
  S->totalCells = scalarOffset;
  /* exa/OptixRenderer.cpp:103-132 gather */
  S->scalars = (float *)xmalloc((size_t)numFields * S->totalCells * sizeof(float));
  S->offsets = (size_t *)xmalloc((size_t)(numFields ? numFields : 1) * sizeof(size_t));
  for (int f = 0; f < numFields; f++) {
    S->offsets[f] = (size_t)f * S->totalCells;
    float *dst = S->scalars + S->offsets[f];
    for (size_t i = 0; i < S->totalCells; i++) {
      int32_t cellID = cellIDs[i];
      if (cellID < 0) {                                           /* :116-121 */
        if (!S->allowEmptyCells) FAIL("overflow in index vector...");
        dst[i] = EMPTY_CELL_POISON_VALUE;                         /* ALLOW_EMPTY_CELLS: :117-118 */
        continue;
      }
      if ((size_t)cellID >= fieldLen[f]) FAIL("invalid cell ID"); /* :125-126 */
      dst[i] = fields[f][cellID];
    }
  }
  /* exa/Regions.cpp:242-320 buildFrom */
  Prim *prims = (Prim *)xmalloc(numBricks * sizeof(Prim));
  float blo[3] = {+INFINITY, +INFINITY, +INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (size_t i = 0; i < numBricks; i++) {
    const OrBrick *b = &S->bricks[i];
    const float cw = (float)(1 << b->level);                     /* Brick::getDomain, Brick.h:50-55 */
    for (int k = 0; k < 3; k++) {
      prims[i].lo[k] = (float)b->lower[k] - 0.5f * cw;
      prims[i].hi[k] = (float)b->lower[k] + ((float)b->size[k] + 0.5f) * cw;
      blo[k] = fminf(blo[k], prims[i].lo[k]);
      bhi[k] = fmaxf(bhi[k], prims[i].hi[k]);
    }
    prims[i].id = (int32_t)i;
  }
  build_rec(S, prims, numBricks, blo, bhi);
  for (size_t r = 0; r < S->numRegions; r++) {                   /* :290-306 */
    OrRegion *R = &S->regions[r];
    int finestLevel = 1 << 30;
    for (int i = 0; i < R->leafListSize; i++) {
      int lv = S->bricks[S->leafList[R->leafListBegin + i]].level;
      if (lv < finestLevel) finestLevel = lv;
    }
    R->finestLevelCellWidth = (float)(1 << finestLevel);
    compute_value_range(S, R, numRegionFields);
  }
  return S;
#undef FAIL
}

void or_scene_destroy(OrScene *S)
{
  if (!S) return;
  free(S->bricks); free(S->scalars); free(S->offsets);
  free(S->regions); free(S->leafList); free(S->nodes);
  free(S->meshVerts); free(S->meshTris); free(S->traces);
  free(S);
}

size_t or_num_bricks(const OrScene *S) { return S->numBricks; }
size_t or_num_regions(const OrScene *S) { return S->numRegions; }
size_t or_leaflist_size(const OrScene *S) { return S->numLeaf; }
size_t or_total_cells(const OrScene *S) { return S->totalCells; }
const OrBrick *or_bricks(const OrScene *S) { return S->bricks; }
const OrRegion *or_regions(const OrScene *S) { return S->regions; }
const int32_t *or_leaflist(const OrScene *S) { return S->leafList; }
const float *or_scalars(const OrScene *S) { return S->scalars; }
void or_voxel_bounds(const OrScene *S, float lo[3], float hi[3])
{ for (int k = 0; k < 3; k++) { lo[k] = S->vb_lo[k]; hi[k] = S->vb_hi[k]; } }

/* OptixRenderer::resetTracer (OptixRenderer.cpp:450-472): seeds at timestep 0, everything else 0, timestep 0 */
void or_reset_tracer(OrScene *S, int enabled, const int channels[3], int numTraces, int numTimesteps, float steplen,
                     const float *seeds)
{
  free(S->traces);
  S->tracerEnabled = enabled; S->numTraces = numTraces; S->numTimesteps = numTimesteps; S->steplen = steplen;
  for (int k = 0; k < 3; k++) S->tracerChannels[k] = channels[k];
  S->traces = (float *)calloc((size_t)numTraces * numTimesteps * 3 + 1, sizeof(float));
  if (!S->traces) die_oom();
  for (int i = 0; i < numTraces; i++) memcpy(&S->traces[(size_t)i * numTimesteps * 3], &seeds[3 * i], 3 * sizeof(float));
  S->timestep = S->timestepHost = 0;
}
/* OptixRenderer::advanceTracer (:474-487): the host counter always advances, the device copy (currentTimestep) only while
 * the counter is <= numTimesteps — further calls leave the picture as it is */
int or_advance_tracer(OrScene *S)
{
  if (!S->tracerEnabled) return 0;
  S->timestepHost++;
  if (S->timestepHost <= S->numTimesteps) { S->timestep = S->timestepHost; return 1; }
  return 0;
}
const float *or_traces(const OrScene *S) { return S->traces; }
int or_tracer_timestep(const OrScene *S) { return S->timestep; }

/* the triangle meshes handed to the OptixRenderer constructor (OptixRenderer.cpp:554-612), world space */
void or_set_triangles(OrScene *S, const float *verts, size_t numVerts, const int32_t *tris, size_t numTris)
{
  free(S->meshVerts); free(S->meshTris);
  S->meshVerts = (float *)xmalloc(numVerts * 3 * sizeof(float));
  S->meshTris = (int32_t *)xmalloc(numTris * 3 * sizeof(int32_t));
  memcpy(S->meshVerts, verts, numVerts * 3 * sizeof(float));
  memcpy(S->meshTris, tris, numTris * 3 * sizeof(int32_t));
  S->numVerts = numVerts; S->numTris = numTris;
}

void or_set_xf(OrScene *S, int chan, const float *rgba128)
{ memcpy(S->xf[chan], rgba128, sizeof(S->xf[chan])); }

/* 1 (default): the tex1D filter weight in 9-bit fixed point with 8 fractional bits, as the CUDA C
 * programming guide publishes it; 0: full-precision weight */
void or_set_tf_filter(OrScene *S, int cudaFixedPoint) { S->tfFracMagic = cudaFixedPoint ? 32768.f : 0.f; }
/* the per-axis association needs "this corner counts" to be a product of per-axis predicates; an empty cell is not, so a
 * scene with empty cells keeps the source order */
void or_set_basis_form(OrScene *S, int form) { S->basisForm = (form && !S->allowEmptyCells) ? 1 : 0; }

/* ------------------------------------------------------------------ */
/* pixel helpers                                                       */
/* ------------------------------------------------------------------ */
/* float -> int as the device does it (CUDA cvt.rzi.s32.f32 and gfx950 v_cvt_i32_f32
 * alike): truncate, saturate, NaN -> 0.  A plain C cast is undefined for those inputs
 * and x86 returns INT_MIN; the reference only ever runs this code on the GPU. */
static inline int f2i(float f)
{
  if (f != f) return 0;
  if (f >= 2147483648.f) return 2147483647;
  if (f <= -2147483648.f) return (-2147483647 - 1);
  return (int)f;
}

/* exabrick.cu:53-60 */
float or_linear_to_srgb(float x)
{
  if (x <= 0.0031308f) return 12.92f * x;
  return 1.055f * powf(x, 1.f / 2.4f) - 0.055f;
}
/* exabrick.cu:62-66 */
int32_t or_make_8bit(float f)
{
  int v = f2i(f * 256.f);
  v = v > 0 ? v : 0;
  return v < 255 ? v : 255;
}
/* exabrick.cu:68-76 */
uint32_t or_make_rgba8(float r, float g, float b)
{
  return ((uint32_t)or_make_8bit(r) << 0) + ((uint32_t)or_make_8bit(g) << 8)
       + ((uint32_t)or_make_8bit(b) << 16) + (0xffu << 24);
}

static inline float clampf(float f, float lo, float hi) { return fminf(hi, fmaxf(lo, f)); }
static inline int clampi(int f, int lo, int hi) { int m = f > lo ? f : lo; return m < hi ? m : hi; }

/* tex1D<float4> on a 128-texel cudaArray, linear filter, clamp addressing,
 * normalized coordinates (exa/Texture.h:141-147; fetch at exabrick.cu:147).  The filter itself
 * runs in NVIDIA's texture unit, outside the reference tree; restated from the CUDA C
 * programming guide, appendix "Texture Fetching", "Linear Filtering":
 *     tex(x) = (1 - a) T[i] + a T[i+1],  i = floor(xB), a = frac(xB), xB = x - 0.5, x = N u
 * "a is stored in 9-bit fixed point format with 8 bits of fractional value (so 1.0 is exactly
 * represented)".  The guide gives the format, not the rounding: the fraction is rounded to the
 * NEAREST multiple of 1/256 here (ties to even) — adding 2^15 to a float in [0,1) leaves exactly 8
 * fractional bits.  fracMagic = 0 keeps the full-precision weight (or_set_tf_filter). */
static inline v4 tex1d_linear(const float (*T)[4], float u, float fracMagic)
{
  float x = u * (float)OR_NUM_XF_VALUES - 0.5f;
  float fl = floorf(x);
  volatile float aq = (x - fl) + fracMagic;      /* volatile: the sum is rounded to binary32 before the subtraction */
  float a = aq - fracMagic;
  int i0 = clampi(f2i(fl), 0, OR_NUM_XF_VALUES - 1);
  int i1 = clampi(f2i(fl) + 1, 0, OR_NUM_XF_VALUES - 1);
  float na = 1.f - a;
  v4 r;
  r.x = na * T[i0][0] + a * T[i1][0];
  r.y = na * T[i0][1] + a * T[i1][1];
  r.z = na * T[i0][2] + a * T[i1][2];
  r.w = na * T[i0][3] + a * T[i1][3];
  return r;
}

/* exabrick.cu:135-150 lookupTransferFunction */
static inline v4 lookup_xf(const OrScene *S, const OrFrameState *fs, float in_scalar, int channel)
{
  float lo = fs->xfDomain[channel][0], hi = fs->xfDomain[channel][1];
  float scalar = (OR_NUM_XF_VALUES - 1) * (in_scalar - lo) / ((hi - lo) + 1e-20f);
  scalar = clampf(scalar + .5f, 0.f, OR_NUM_XF_VALUES - 1.f);
  scalar /= OR_NUM_XF_VALUES - 1.f;
  v4 r = tex1d_linear(S->xf[channel], scalar, S->tfFracMagic);
  r.w *= fs->xfOpacityScale;
  return r;
}
void or_lookup_xf(const OrScene *S, const OrFrameState *fs, float v, int chan, float rgba[4])
{ v4 r = lookup_xf(S, fs, v, chan); rgba[0] = r.x; rgba[1] = r.y; rgba[2] = r.z; rgba[3] = r.w; }

typedef struct { v3 org, dir; float tmin, tmax; } Ray;

/* exabrick.cu:197-210 boxTest (true division; fminf/fmaxf NaN-ignoring) */
static inline int box_test(const Ray *ray, const float lo[3], const float hi[3], float *t0, float *t1)
{
  v3 t_lo = vdiv(vsub(vfrom(lo), ray->org), ray->dir);
  v3 t_hi = vdiv(vsub(vfrom(hi), ray->org), ray->dir);
  v3 t_nr = V3(fminf(t_lo.x, t_hi.x), fminf(t_lo.y, t_hi.y), fminf(t_lo.z, t_hi.z));
  v3 t_fr = V3(fmaxf(t_lo.x, t_hi.x), fmaxf(t_lo.y, t_hi.y), fmaxf(t_lo.z, t_hi.z));
  *t0 = fmaxf(ray->tmin, fmaxf(fmaxf(t_nr.x, t_nr.y), t_nr.z));
  *t1 = fminf(ray->tmax, fminf(fminf(t_fr.x, t_fr.y), t_fr.z));
  return *t0 < *t1;
}
int or_box_test(const float org[3], const float dir[3], float tmin, float tmax,
                const float lo[3], const float hi[3], float *t0, float *t1)
{ Ray r = {vfrom(org), vfrom(dir), tmin, tmax}; return box_test(&r, lo, hi, t0, t1); }

/* exabrick.cu:250-281 activeForVolumeSampling */
static int active_for_volume_sampling(const OrScene *S, const OrFrameState *fs,
                                      float vlo, float vhi, int channel)
{
  float dlo = fs->xfDomain[channel][0], dhi = fs->xfDomain[channel][1];
  if (vlo > dhi) return 0;
  if (vhi < dlo) return 0;
  const float scaled_lo = (vlo - dlo) / ((dhi - dlo) + 1e-20f);
  const float scaled_hi = (vhi - dlo) / ((dhi - dlo) + 1e-20f);
  const int idx_lo = clampi(f2i(scaled_lo * (OR_NUM_XF_VALUES - 1)), 0, OR_NUM_XF_VALUES - 1);
  const int idx_hi = clampi(f2i(scaled_hi * (OR_NUM_XF_VALUES - 1)) + 1, 0, OR_NUM_XF_VALUES - 1);
  for (int i = idx_lo; i <= idx_hi; i++) {
    float cellValue = (float)i / (OR_NUM_XF_VALUES - 1);
    cellValue *= dhi - dlo;
    cellValue += dlo;
    v4 rgba = lookup_xf(S, fs, cellValue, channel);
    if (rgba.w > 0.f) return 1;
  }
  return 0;
}

/* exabrick.cu:285-312 VolumeBVH bounds program: which regions get a real box */
void or_volume_active(const OrScene *S, const OrFrameState *fs, const OrParams *P, uint8_t *active)
{
  for (size_t r = 0; r < S->numRegions; r++) {
    int a = 0;
    for (int c = 0; c < P->numChannels; ++c) {
      a |= active_for_volume_sampling(S, fs, S->regions[r].vr_lo, S->regions[r].vr_hi, c);
      if (a) break;
    }
    active[r] = (uint8_t)(P->spaceSkippingEnabled ? a : 1);
  }
}
/* exabrick.cu:373-402 IsoSurface bounds program */
void or_iso_active(const OrScene *S, const OrFrameState *fs, uint8_t *active)
{
  for (size_t r = 0; r < S->numRegions; r++) {
    int a = 0;
    for (int i = 0; i < OR_MAX_ISO_SURFACES; i++)
      if (fs->iso[i].enabled && fs->iso[i].value >= S->regions[r].vr_lo
          && fs->iso[i].value <= S->regions[r].vr_hi) a = 1;
    active[r] = (uint8_t)a;
  }
}

/* ------------------------------------------------------------------ */
/* closest-region query.  Semantics of exabrick.cu:184-238: among the   */
/* regions with a real box, the one with the smallest clamped entry t0  */
/* (t0>=ray.tmin, t0<t1).  OptiX's tie-break at equal t0 is not         */
/* observable; the oracle takes the lowest region id.  t1 is clamped to */
/* the ray's own tmax only.  The kd tree recorded by      */
/* build_rec is used purely to prune; result == brute force.            */
/* ------------------------------------------------------------------ */
typedef struct { int leafID; float t0, t1; } RegionHit;

static RegionHit trace_brute(const OrScene *S, const uint8_t *active, const Ray *ray)
{
  RegionHit h = {-1, 0.f, 0.f};
  for (size_t r = 0; r < S->numRegions; r++) {
    if (!active[r]) continue;
    float t0, t1;
    if (!box_test(ray, S->regions[r].dom_lo, S->regions[r].dom_hi, &t0, &t1)) continue;
    if (h.leafID < 0 || t0 < h.t0) { h.leafID = (int)r; h.t0 = t0; h.t1 = t1; }
  }
  return h;
}

static RegionHit trace_kd(const OrScene *S, const uint8_t *active, const Ray *ray)
{
  RegionHit h = {-1, 0.f, 0.f};
  if (S->numNodes == 0) return h;
  int32_t stack[256];
  int sp = 0;
  stack[sp++] = 0;
  while (sp) {
    const KdNode *N = &S->nodes[stack[--sp]];
    float t0, t1;
    box_test(ray, N->lo, N->hi, &t0, &t1);
    if (!(t0 <= t1)) continue;                       /* conservative: keep touching boxes */
    if (h.leafID >= 0 && t0 > h.t0) continue;        /* cannot beat (or tie) the best   */
    if (N->left < 0 && N->right < 0) {
      int r = N->region;
      if (r < 0 || !active[r]) continue;
      if (!(t0 < t1)) continue;
      if (h.leafID < 0 || t0 < h.t0 || (t0 == h.t0 && r < h.leafID)) { h.leafID = r; h.t0 = t0; h.t1 = t1; }
      continue;
    }
    /* push far child first so the near child is popped first (speed only) */
    int32_t a = N->left, b = N->right;
    if (a >= 0 && b >= 0) {
      /* split axis = the axis on which the children differ */
      const KdNode *L = &S->nodes[a];
      int ax = 0;
      for (int k = 0; k < 3; k++) if (L->hi[k] != N->hi[k]) ax = k;
      if (vget(ray->dir, ax) < 0.f) { int32_t t = a; a = b; b = t; }
      if (sp + 2 > 256) { fprintf(stderr, "exa_oracle: kd stack overflow\n"); abort(); }
      stack[sp++] = b; stack[sp++] = a;
    } else {
      if (sp + 1 > 256) { fprintf(stderr, "exa_oracle: kd stack overflow\n"); abort(); }
      stack[sp++] = a >= 0 ? a : b;
    }
  }
  return h;
}

static inline RegionHit trace_region(const OrScene *S, const uint8_t *active, const Ray *ray)
{
  return S->numRegions <= 64 ? trace_brute(S, active, ray) : trace_kd(S, active, ray);
}

int or_trace_region(const OrScene *S, const uint8_t *active, const float org[3],
                    const float dir[3], float tmin, float tmax, float *t0, float *t1)
{
  Ray r = {vfrom(org), vfrom(dir), tmin, tmax};
  RegionHit a = trace_brute(S, active, &r), b = trace_kd(S, active, &r);
  if (a.leafID != b.leafID || (a.leafID >= 0 && (a.t0 != b.t0 || a.t1 != b.t1))) {
    fprintf(stderr, "exa_oracle: kd/brute mismatch %d(%g,%g) vs %d(%g,%g)\n",
            a.leafID, a.t0, a.t1, b.leafID, b.t0, b.t1);
    return -2;
  }
  if (a.leafID >= 0) { *t0 = a.t0; *t1 = a.t1; }
  return a.leafID;
}

/* ------------------------------------------------------------------ */
/* basis-function reconstruction                                       */
/* ------------------------------------------------------------------ */
typedef struct {
  const OrScene *S; const OrFrameState *fs; const OrParams *P;
  const uint8_t *volActive, *isoActive;
  OrStats st;
} Ctx;

/* exabrick.cu:581-594 getScalar (reference adds in 32-bit unsigned; the oracle
 * uses size_t — identical whenever the reference does not overflow) */
static inline float get_scalar(Ctx *C, const OrBrick *b, int ix, int iy, int iz, int channel)
{
  size_t idx = (size_t)b->begin + (size_t)ix + (size_t)iy * b->size[0] + (size_t)iz * b->size[0] * b->size[1];
  C->st.corner_loads++;
  return C->S->scalars[C->S->offsets[channel] + idx];
}

typedef struct { float sumWV, sumW; v3 sumD, sumDC; } Basis;

/* exabrick.cu:596-612 add() overloads */
static inline void add1(float *sumWeights, float *sumWeightedValues, float weight, float scalar)
{ *sumWeights += weight; *sumWeightedValues += weight * scalar; }
static inline void add3(v3 *sumWeights, v3 *sumWeightedValues, v3 weight, float scalar)
{
  *sumWeights = vadd(*sumWeights, weight);
  *sumWeightedValues = vadd(*sumWeightedValues, vmul(weight, v3s(scalar)));
}

/* exabrick.cu:620-777 addBasisFunctions<NEED_DERIVATIVE> (INV_CELL_WIDTH == 1.f, :641) */
static void add_basis_functions(Ctx *C, Basis *B, int need_derivative, int brickID, v3 pos, int channel)
{
  const OrBrick *brick = &C->S->bricks[brickID];
  const float cellWidth = (float)(1 << brick->level);
  C->st.brick_visits++;

  const v3 lower = V3((float)brick->lower[0], (float)brick->lower[1], (float)brick->lower[2]);
  const v3 localPos = vsub(vdiv(vsub(pos, lower), v3s(cellWidth)), v3s(0.5f));
  int lx = f2i(floorf(localPos.x)), ly = f2i(floorf(localPos.y)), lz = f2i(floorf(localPos.z));
  lx = lx > -1 ? lx : -1; ly = ly > -1 ? ly : -1; lz = lz > -1 ? lz : -1;   /* max(vec3i(-1),idx_lo) */
  const int hx = lx + 1, hy = ly + 1, hz = lz + 1;
  const v3 frac = vsub(localPos, V3((float)lx, (float)ly, (float)lz));
  const v3 neg_frac = vsub(v3s(1.f), frac);
  const int sx = brick->size[0], sy = brick->size[1], sz = brick->size[2];
#define CORNER(IX, IY, IZ, WZ, WY, WX, SDX, SDY, SDZ)                                        \
  do {                                                                                       \
    const float scalar = get_scalar(C, brick, IX, IY, IZ, channel);                          \
    if (C->S->allowEmptyCells && !(scalar != EMPTY_CELL_POISON_VALUE)) break;   /* notEmptyCell, :614-618, :646 ... */ \
    const float weight = (WZ) * (WY) * (WX);                                                 \
    if (need_derivative) {                                                                   \
      const float dx = (WZ) * (WY) * (SDX 1.f);                                              \
      const float dy = (WZ) * (WX) * (SDY 1.f);                                              \
      const float dz = (WY) * (WX) * (SDZ 1.f);                                              \
      add3(&B->sumDC, &B->sumD, V3(dx, dy, dz), scalar);                                     \
    }                                                                                        \
    add1(&B->sumW, &B->sumWV, weight, scalar);                                               \
  } while (0)
  if (lz >= 0 && lz < sz) {
    if (ly >= 0 && ly < sy) {
      if (lx >= 0 && lx < sx) CORNER(lx, ly, lz, neg_frac.z, neg_frac.y, neg_frac.x, -, -, -); /* :644-658 */
      if (hx < sx)            CORNER(hx, ly, lz, neg_frac.z, neg_frac.y, frac.x,     +, -, -); /* :659-673 */
    }
    if (hy < sy) {
      if (lx >= 0 && lx < sx) CORNER(lx, hy, lz, neg_frac.z, frac.y, neg_frac.x,     -, +, -); /* :676-691 */
      if (hx < sx)            CORNER(hx, hy, lz, neg_frac.z, frac.y, frac.x,         +, +, -); /* :692-706 */
    }
  }
  if (hz < sz) {
    if (ly >= 0 && ly < sy) {
      if (lx >= 0 && lx < sx) CORNER(lx, ly, hz, frac.z, neg_frac.y, neg_frac.x,     -, -, +); /* :712-726 */
      if (hx < sx)            CORNER(hx, ly, hz, frac.z, neg_frac.y, frac.x,         +, -, +); /* :727-741 */
    }
    if (hy < sy) {
      if (lx >= 0 && lx < sx) CORNER(lx, hy, hz, frac.z, frac.y, neg_frac.x,         -, +, +); /* :744-758 */
      if (hx < sx)            CORNER(hx, hy, hz, frac.z, frac.y, frac.x,             +, +, +); /* :759-774 */
    }
  }
#undef CORNER
}

/* Second association of the same sums (or_set_basis_form(S, 1)).
 *
 * The reference's source adds the eight corners one after the other, each weight built as (wz*wy)*wx (:644-774).  The
 * reference BINARY does not compute that sequence either: nvcc contracts a*b+c into fused multiply-adds by default and
 * CMakeLists.txt passes no -fmad=false, so which roundings the shipped renderer performs is a property of nvcc's
 * instruction selection.  Both sequences are therefore restatements of the same real-valued expression
 *     sumWV += SUM_zyx wz*wy*wx*s_zyx        sumW  += SUM_zyx wz*wy*wx
 *     sumD  += SUM_zyx grad(wz*wy*wx)*s_zyx  sumDC += SUM_zyx grad(wz*wy*wx)
 * over the corners that lie inside the brick.  "Inside" is a product of per-axis predicates, so the per-axis weights of
 * an outside corner are set to 0 (and its value is not read: it counts as +0) and the triple sum factors:
 *     row (z,y):  a = wxl*s_l + wxh*s_h                d = mxl*s_l + mxh*s_h         (mx = d wx / dx = -1 | +1 | 0)
 *     y:          A_z = wyl*a_zl + wyh*a_zh            DX_z = wyl*d_zl + wyh*d_zh    DY_z = myl*a_zl + myh*a_zh
 *     z:          sumWV += wzl*A_l + wzh*A_h           sumD.x += wzl*DX_l + wzh*DX_h sumD.y += ... DY   sumD.z += mzl*A_l + mzh*A_h
 *     weights:    sumW += (Sz*Sy)*Sx    sumDC.x += (Sz*Sy)*Mx    sumDC.y += (Sz*Sx)*My    sumDC.z += (Sy*Sx)*Mz
 * with S = wl + wh and M = ml + mh per axis.  Every `p*q + r` below is ONE fused multiply-add (fmaf), every other
 * operation one IEEE binary32 operation, in exactly this order; the HIP kernels execute the same sequence
 * (exa_kernels.hip: addBasisFast under EXA_BASIS_FORM == 1).  49 operations per brick with derivatives where the source
 * order takes 116.  Form 1 also fuses the multiply-adds AROUND these sums, each at its site below: the cell coordinate
 * (pos - lower) / cw - 0.5 here, the gradient sumW * sumD - sumWV * sumDC (sample_point_with_derivative), the DVR sample
 * position org + t * dir (integrate_brick), the three dot products of the shading factor and the colour terms of the "over"
 * operator (integrate_volume). */
static void add_basis_functions_factored(Ctx *C, Basis *B, int need_derivative, int brickID, v3 pos, int channel)
{
  const OrBrick *brick = &C->S->bricks[brickID];
  const float cellWidth = (float)(1 << brick->level);
  C->st.brick_visits++;

  /* position in the brick (:624-632) with the division by the power-of-two cell width as an exact multiplication fused with
   * the -0.5; low cell index and fraction as in the source order */
  const v3 lower = V3((float)brick->lower[0], (float)brick->lower[1], (float)brick->lower[2]);
  const float invCw = 1.f / cellWidth;
  const v3 localPos = V3(fmaf(pos.x - lower.x, invCw, -0.5f), fmaf(pos.y - lower.y, invCw, -0.5f), fmaf(pos.z - lower.z, invCw, -0.5f));
  int lx = f2i(floorf(localPos.x)), ly = f2i(floorf(localPos.y)), lz = f2i(floorf(localPos.z));
  lx = lx > -1 ? lx : -1; ly = ly > -1 ? ly : -1; lz = lz > -1 ? lz : -1;
  const int hx = lx + 1, hy = ly + 1, hz = lz + 1;
  const v3 frac = vsub(localPos, V3((float)lx, (float)ly, (float)lz));
  const v3 neg_frac = vsub(v3s(1.f), frac);
  const int sx = brick->size[0], sy = brick->size[1], sz = brick->size[2];
  /* the reference's in-brick tests, per axis (:642-643, :659, ...) */
  const int vlx = lx >= 0 && lx < sx, vhx = hx < sx;
  const int vly = ly >= 0 && ly < sy, vhy = hy < sy;
  const int vlz = lz >= 0 && lz < sz, vhz = hz < sz;
  const float wxl = vlx ? neg_frac.x : 0.f, wxh = vhx ? frac.x : 0.f;
  const float wyl = vly ? neg_frac.y : 0.f, wyh = vhy ? frac.y : 0.f;
  const float wzl = vlz ? neg_frac.z : 0.f, wzh = vhz ? frac.z : 0.f;
  const float mxl = vlx ? -1.f : 0.f, mxh = vhx ? 1.f : 0.f;
  const float myl = vly ? -1.f : 0.f, myh = vhy ? 1.f : 0.f;
  const float mzl = vlz ? -1.f : 0.f, mzh = vhz ? 1.f : 0.f;
  /* cell values; a corner outside the brick is not read */
  float s[2][2][2];
  for (int iz = 0; iz < 2; iz++)
    for (int iy = 0; iy < 2; iy++)
      for (int ix = 0; ix < 2; ix++) {
        const int ok = (ix ? vhx : vlx) && (iy ? vhy : vly) && (iz ? vhz : vlz);
        s[iz][iy][ix] = ok ? get_scalar(C, brick, ix ? hx : lx, iy ? hy : ly, iz ? hz : lz, channel) : 0.f;
      }
  float A[2], DX[2], DY[2];
  for (int iz = 0; iz < 2; iz++) {
    float a[2], d[2];
    for (int iy = 0; iy < 2; iy++) {
      a[iy] = fmaf(wxh, s[iz][iy][1], wxl * s[iz][iy][0]);
      if (need_derivative) d[iy] = fmaf(mxh, s[iz][iy][1], mxl * s[iz][iy][0]);
    }
    A[iz] = fmaf(wyh, a[1], wyl * a[0]);
    if (need_derivative) {
      DX[iz] = fmaf(wyh, d[1], wyl * d[0]);
      DY[iz] = fmaf(myh, a[1], myl * a[0]);
    }
  }
  B->sumWV = fmaf(wzh, A[1], fmaf(wzl, A[0], B->sumWV));
  const float Sx = wxl + wxh, Sy = wyl + wyh, Sz = wzl + wzh;
  const float zy = Sz * Sy;
  B->sumW = fmaf(zy, Sx, B->sumW);
  if (need_derivative) {
    B->sumD.x = fmaf(wzh, DX[1], fmaf(wzl, DX[0], B->sumD.x));
    B->sumD.y = fmaf(wzh, DY[1], fmaf(wzl, DY[0], B->sumD.y));
    B->sumD.z = fmaf(mzh, A[1], fmaf(mzl, A[0], B->sumD.z));
    const float Mx = mxl + mxh, My = myl + myh, Mz = mzl + mzh;
    B->sumDC.x = fmaf(zy, Mx, B->sumDC.x);
    B->sumDC.y = fmaf(Sz * Sx, My, B->sumDC.y);
    B->sumDC.z = fmaf(Sy * Sx, Mz, B->sumDC.z);
  }
}

static inline void add_basis(Ctx *C, Basis *B, int need_derivative, int brickID, v3 pos, int channel)
{
  if (C->S->basisForm) add_basis_functions_factored(C, B, need_derivative, brickID, pos, channel);
  else                 add_basis_functions(C, B, need_derivative, brickID, pos, channel);
}

/* exabrick.cu:781-806 samplePoint */
static int sample_point(Ctx *C, float *value, int leafID, v3 pos, int channel)
{
  const OrRegion *region = &C->S->regions[leafID];
  const int32_t *childList = &C->S->leafList[region->leafListBegin];
  Basis B; memset(&B, 0, sizeof(B));
  for (int childID = 0; childID < region->leafListSize; childID++)
    add_basis(C, &B, 0, childList[childID], pos, channel);
  if (B.sumW <= 1e-20f) return 0;
  *value = B.sumWV / B.sumW;
  return 1;
}

/* exabrick.cu:883-928 samplePointWithDerivative (ANALYTIC_GRADIENTS=1) */
static int sample_point_with_derivative(Ctx *C, float *value, v3 *derivatives, int leafID, v3 pos, int channel)
{
  const OrRegion *region = &C->S->regions[leafID];
  const int32_t *childList = &C->S->leafList[region->leafListBegin];
  Basis B; memset(&B, 0, sizeof(B));
  for (int childID = 0; childID < region->leafListSize; childID++)
    add_basis(C, &B, 1, childList[childID], pos, channel);
  if (B.sumW <= 1e-20f) return 0;
  *value = B.sumWV / B.sumW;
  if (C->S->basisForm)      /* form 1: the product-difference with its first product fused */
    *derivatives = V3(fmaf(B.sumW, B.sumD.x, -(B.sumWV * B.sumDC.x)),
                      fmaf(B.sumW, B.sumD.y, -(B.sumWV * B.sumDC.y)),
                      fmaf(B.sumW, B.sumD.z, -(B.sumWV * B.sumDC.z)));
  else
  *derivatives = V3(B.sumW * B.sumD.x - B.sumWV * B.sumDC.x,
                    B.sumW * B.sumD.y - B.sumWV * B.sumDC.y,
                    B.sumW * B.sumD.z - B.sumWV * B.sumDC.z);       /* :916-918 */
  return 1;
}

int or_sample_point(const OrScene *S, int regionID, const float pos[3], int chan,
                    int withDerivative, float *value, float grad[3])
{
  Ctx C; memset(&C, 0, sizeof(C)); C.S = S;
  v3 g = v3s(0.f);
  int ok = withDerivative ? sample_point_with_derivative(&C, value, &g, regionID, vfrom(pos), chan)
                          : sample_point(&C, value, regionID, vfrom(pos), chan);
  if (grad) { grad[0] = g.x; grad[1] = g.y; grad[2] = g.z; }
  return ok;
}

/* ------------------------------------------------------------------ */
/* integration                                                         */
/* ------------------------------------------------------------------ */
typedef struct { v4 *pixelColor; float t_hit; v3 gradient; } IntegrationResult; /* :967-986 */

/* exabrick.cu:988-1016 integrateVolume */
static void integrate_volume(Ctx *C, const Ray *ray, IntegrationResult *result, float actual_dt,
                             float cellValue, v3 gradient, int finestLevelCellWidth, int channel)
{
  if (actual_dt == 0.f) return;
  v4 *pixelColor = result->pixelColor;
  v4 sample = lookup_xf(C->S, C->fs, cellValue, channel);
  if (C->S->basisForm) {
    /* form 1: the same expressions with the additions of the three dot products and of the "over" operator's colour terms
     * fused into their products (vdotf: left to right, like vdot) */
    if (sqrtf(vdotf(gradient, gradient)) > finestLevelCellWidth * 1e-6f) {
      const v3 lightDir = vneg(ray->dir);
      const float scale = fabsf(vdotf(lightDir, gradient))
                        / sqrtf(vdotf(gradient, gradient) * vdotf(lightDir, lightDir));
      sample.x *= scale; sample.y *= scale; sample.z *= scale;
    }
    sample.w = 1.f - powf(1.f - sample.w, actual_dt);
    const float k = (1.f - pixelColor->w) * sample.w;
    pixelColor->x = fmaf(k, sample.x, pixelColor->x);
    pixelColor->y = fmaf(k, sample.y, pixelColor->y);
    pixelColor->z = fmaf(k, sample.z, pixelColor->z);
    pixelColor->w += k * 1.f;
    return;
  }
  if (vlength(gradient) > finestLevelCellWidth * 1e-6f) {
    const v3 lightDir = vneg(ray->dir);
    const float scale = fabsf(vdot(lightDir, gradient))
                      / sqrtf(vdot(gradient, gradient) * vdot(lightDir, lightDir));
    sample.x *= scale; sample.y *= scale; sample.z *= scale;
  }
  sample.w = 1.f - powf(1.f - sample.w, actual_dt);
  const float k = (1.f - pixelColor->w) * sample.w;             /* :1012 */
  pixelColor->x += k * sample.x;
  pixelColor->y += k * sample.y;
  pixelColor->z += k * sample.z;
  pixelColor->w += k * 1.f;
}

/* exabrick.cu:1018-1114 IsoSurfaceIntegrationFunction */
typedef struct { float last_t, lastCellValue; v3 lastGradient; } IsoFunc;

static void iso_func_call(Ctx *C, IsoFunc *F, const Ray *ray, IntegrationResult *result,
                          float t_sample, float cellValueIn, v3 gradient, int leafID, int channel)
{
  const OrFrameState *fs = C->fs;
  if (F->lastCellValue >= -1e35f) {
    for (int i = 0; i < OR_MAX_ISO_SURFACES; i++) {
      const float isoV = fs->iso[i].value;
      if (fs->iso[i].enabled && fs->iso[i].channel == channel
          && ((F->lastCellValue <= isoV && cellValueIn >= isoV)
              || (F->lastCellValue >= isoV && cellValueIn <= isoV))) {
        float iso = isoV;
        float d1 = fabsf(F->lastCellValue - iso);
        float d2 = fabsf(cellValueIn - iso);
        float w1 = 1.f - d1 / (d1 + d2);
        float w2 = 1.f - d2 / (d1 + d2);
        float tavg = F->last_t * w1 + t_sample * w2;              /* :1053 */
        float cellValue = 0.f;
        v3 grad = v3s(0.f);
        v4 sample = {1.f, 0.f, 0.f, 1.f};
        const v3 isopt = vadd(ray->org, vscale(tavg, ray->dir));

        if (C->P->gradientShadingISO) {
          C->st.iso_evals++;
          if (sample_point_with_derivative(C, &cellValue, &grad, leafID, isopt, fs->iso[i].channel)) {
            sample = lookup_xf(C->S, fs, cellValue, fs->iso[i].channel);
            grad = vnormalize(grad);
            if (vdot(grad, ray->dir) > 0.f) grad = vneg(grad);    /* :1068-1070 */
          }
        } else {
          C->st.iso_evals++;
          if (sample_point(C, &cellValue, leafID, isopt, fs->iso[i].channel))
            sample = lookup_xf(C->S, fs, cellValue, fs->iso[i].channel);
        }
        if (C->P->colormapChannel != 0) {                         /* :1079-1085 */
          cellValue = 0.f;
          C->st.iso_evals++;
          if (sample_point(C, &cellValue, leafID, isopt, C->P->colormapChannel))
            sample = lookup_xf(C->S, fs, cellValue, 0);
        }
        sample.w = 1.f;
        if (!isfinite(grad.x) || !isfinite(grad.y) || !isfinite(grad.z)) grad = v3s(0.f);
        if (vlength(grad) > .0f) {
          const v3 lightDir = vneg(ray->dir);
          const float scale = .3f + .7f * fabsf(vdot(lightDir, grad)) / sqrtf(vdot(grad, grad));
          sample.x *= scale; sample.y *= scale; sample.z *= scale;
        }
        v4 *pc = result->pixelColor;
        const float k = (1.f - pc->w) * sample.w;                 /* :1099 */
        pc->x += k * sample.x; pc->y += k * sample.y; pc->z += k * sample.z; pc->w += k * 1.f;
        result->t_hit = tavg;
        result->gradient = grad;
      }
    }
  }
  F->last_t = t_sample;
  F->lastCellValue = cellValueIn;
  F->lastGradient = gradient;
}

#define TERMINATION_THRESHOLD 0.98f /* exabrick.cu:49 */

/* first sample position, exabrick.cu:1141-1144 */
static inline float first_t(float t0, float dt, float off)
{
  int i0 = f2i(ceilf((t0 - dt * off) / dt));
  float t_i = (off + i0) * dt;
  while ((t_i - dt) >= t0) t_i = t_i - dt;
  while (t_i < t0) t_i += dt;
  return t_i;
}

/* exabrick.cu:1116-1185 integrateBrick<GRADIENT_SHADING> */
static void integrate_brick(Ctx *C, int gradient_shading, IntegrationResult *result, float off,
                            const Ray *ray, int leafID, float t0, float t1, int numChannels)
{
  const OrRegion *region = &C->S->regions[leafID];
  const float dt = C->P->dt * region->finestLevelCellWidth;
  const int finestLevelCellWidth = f2i(region->finestLevelCellWidth);
  float t_i = first_t(t0, dt, off);
  float t_last = t0;
  for (;; t_i += dt) {
    const float t_next = fminf(t_i, t1);
    const float t_sample = 0.5f * (fminf(t1, t_next) + t_last);
    const float actual_dt = t_next - t_last;
    t_last = t_next;
    /* form 1: org + t * dir with the addition fused into the product (the DVR march only; the iso march keeps two roundings) */
    const v3 pos = C->S->basisForm ? V3(fmaf(t_sample, ray->dir.x, ray->org.x), fmaf(t_sample, ray->dir.y, ray->org.y),
                                        fmaf(t_sample, ray->dir.z, ray->org.z))
                                   : vadd(ray->org, vscale(t_sample, ray->dir));
    float cellValue = 0.f;
    v3 grad = v3s(0.f);
    for (int c = 0; c < numChannels; ++c) {
      C->st.sample_evals++;
      int ok = gradient_shading ? sample_point_with_derivative(C, &cellValue, &grad, leafID, pos, c)
                                : sample_point(C, &cellValue, leafID, pos, c);
      if (ok) {
        C->st.samples++;
        integrate_volume(C, ray, result, actual_dt, cellValue, grad, finestLevelCellWidth, c);
      }
    }
    if (result->pixelColor->w >= TERMINATION_THRESHOLD) break;
    if (t_next >= t1) break;
  }
}

/* exabrick.cu:1187-1256 isoIntegrateBrick */
static void iso_integrate_brick(Ctx *C, IsoFunc *funcs, IntegrationResult *result, float off,
                                const Ray *ray, int leafID, float t0, float t1, int numChannels)
{
  const OrRegion *region = &C->S->regions[leafID];
  const float dt = C->P->dt * region->finestLevelCellWidth;
  float t_i = first_t(t0, dt, off);
  float t_last = t0;
  for (;; t_i += dt) {
    const float t_next = fminf(t_i, t1);
    const float t_sample = 0.5f * (fminf(t1, t_next) + t_last);
    t_last = t_next;
    const v3 pos = vadd(ray->org, vscale(t_sample, ray->dir));
    for (int c = 0; c < numChannels; ++c) {
      int doIntegrate;
      float cellValue = 0.f;
      v3 grad = v3s(0.f);
      C->st.iso_evals++;
      if (C->P->gradientShadingISO)
        doIntegrate = sample_point_with_derivative(C, &cellValue, &grad, leafID, pos, c);
      else
        doIntegrate = sample_point(C, &cellValue, leafID, pos, c);
      if (doIntegrate) {
        iso_func_call(C, &funcs[c], ray, result, t_sample, cellValue, grad, leafID, c);
        if (result->pixelColor->w >= TERMINATION_THRESHOLD) break;   /* leaves the channel loop only */
      }
    }
    if (t_next >= t1) break;
  }
}

/* exabrick.cu:412-418 SurfacePRD */
#define PRIMID_ISOSURFACE (-23)
typedef struct { int primID; float t_hit; v3 Ng; float ambient; v3 baseColor; } SurfacePRD;

/* exabrick.cu:1408-1460 traceIsoRay */
static SurfacePRD trace_iso_ray(Ctx *C, Ray ray, float off)
{
  const OrFrameState *fs = C->fs;
  ray.org = xfm_point(fs, ray.org);
  ray.dir = xfm_vector(fs, ray.dir);
  const float dt_scale = vlength(ray.dir);
  ray.dir = vnormalize(ray.dir);
  float alreadyIntegratedDistance = dt_scale * ray.tmin;
  IsoFunc funcs[OR_MAX_CHANNELS];
  for (int c = 0; c < OR_MAX_CHANNELS; c++) {               /* :1019-1022 */
    funcs[c].lastCellValue = -1e36f; funcs[c].lastGradient = v3s(0.f); funcs[c].last_t = 0.f;
  }
  SurfacePRD result; memset(&result, 0, sizeof(result));
  result.primID = -1;   /* the reference leaves it uninitialised; callers test only for == -23 */
  result.t_hit = ray.tmax;
  for (;;) {
    ray.tmin = alreadyIntegratedDistance;
    ray.tmax = ray.tmax * dt_scale;                           /* :1434, re-applied every iteration */
    RegionHit prd = trace_region(C->S, C->isoActive, &ray);
    if (prd.leafID < 0) break;
    C->st.iso_segments++;
    v4 pixelColor = {0.f, 0.f, 0.f, 0.f};
    IntegrationResult ir = {&pixelColor, -1.f, {0.f, 0.f, 0.f}};
    iso_integrate_brick(C, funcs, &ir, off, &ray, prd.leafID,
                        fmaxf(ray.tmin, prd.t0), fminf(ray.tmax, prd.t1), C->P->numPrimaryChannels);
    if (ir.t_hit >= 0.f) {
      result.primID = PRIMID_ISOSURFACE;
      result.t_hit = ir.t_hit / dt_scale;
      result.Ng = vnormalize(ir.gradient);
      result.ambient = 0.f;
      result.baseColor = V3(pixelColor.x, pixelColor.y, pixelColor.z);
      return result;
    }
    alreadyIntegratedDistance = prd.t1 * (1.0000001f);
  }
  return result;
}

/* ------------------------------------------------------------------ */
/* contour planes                                   */
/* ------------------------------------------------------------------ */
#define PRIMID_PLANE (-24)

/* exabrick.cu:1267-1284 */
static float intersect_line_plane(v3 p1, v3 p2, v3 normal, float offset)
{
  float s = vdot(normal, vnormalize(vsub(p2, p1)));
  if (s == 0.f) return -1.f;
  float t = (offset - vdot(normal, p1)) / s;
  if (t < 0.f || t > vlength(vsub(p2, p1))) return -1.f;
  return t;
}

/* exabrick.cu:1287-1314 */
static void intersect_box_plane(v3 blo, v3 bhi, v3 normal, float offset, v3 *pts, int *isectCnt)
{
  static const int key[4][3] = { {0,0,0}, {1,0,1}, {1,1,0}, {0,1,1} };
  const v3 corners[2] = { blo, bhi };
  *isectCnt = 0;
  for (int i = 0; i < 4 && *isectCnt < 6; ++i) {
    for (int j = 0; j < 3 && *isectCnt < 6; ++j) {
      v3 p1 = V3((j == 0) ? corners[1 - key[i][0]].x : corners[key[i][0]].x,
                 (j == 1) ? corners[1 - key[i][1]].y : corners[key[i][1]].y,
                 (j == 2) ? corners[1 - key[i][2]].z : corners[key[i][2]].z);
      v3 p2 = V3(corners[key[i][0]].x, corners[key[i][1]].y, corners[key[i][2]].z);
      float t = intersect_line_plane(p1, p2, normal, offset);
      if (t >= 0.f) pts[(*isectCnt)++] = vadd(p1, vscale(t, vnormalize(vsub(p2, p1))));
    }
  }
}

/* exabrick.cu:1316-1343 */
static float intersect_ray_triangle(const Ray *ray, v3 v1, v3 e1, v3 e2)
{
  v3 s1 = vcross(ray->dir, e2);
  float div = vdot(s1, e1);
  if (div == 0.f) return -1.f;
  float invDiv = 1.f / div;
  v3 d = vsub(ray->org, v1);
  float b1 = vdot(d, s1) * invDiv;
  if (b1 < 0.f || b1 > 1.f) return -1.f;
  v3 s2 = vcross(d, e1);
  float b2 = vdot(ray->dir, s2) * invDiv;
  if (b2 < 0.f || b1 + b2 > 1.f) return -1.f;
  return vdot(e2, s2) * invDiv;
}

/* exabrick.cu:818-830 samplePointWithInfRay.  The reference indexes region[-1] when the
 * degenerate ray hits nothing (undefined); here that case yields 0 and the sample is skipped. */
static float sample_point_with_inf_ray(Ctx *C, v3 pos, int channel)
{
  Ray ray = { pos, V3(1.f, 1.f, 1.f), 0.f, 2e-10f };
  RegionHit prd = trace_region(C->S, C->volActive, &ray);
  float value = 0.f;
  if (prd.leafID >= 0) sample_point(C, &value, prd.leafID, pos, channel);
  return value;
}

/* rcp(affine3f) of the un-vendored owl: inverse of the linear part by adjoint/determinant */
static void world_space_bounds(const OrScene *S, const OrFrameState *fs, v3 *wlo, v3 *whi)
{
  v3 vx = vfrom(fs->xfm_vx), vy = vfrom(fs->xfm_vy), vz = vfrom(fs->xfm_vz), p = vfrom(fs->xfm_p);
  v3 c0 = vcross(vy, vz), c1 = vcross(vz, vx), c2 = vcross(vx, vy);
  float det = vdot(vx, c0);
  v3 ix = V3(c0.x / det, c1.x / det, c2.x / det);
  v3 iy = V3(c0.y / det, c1.y / det, c2.y / det);
  v3 iz = V3(c0.z / det, c1.z / det, c2.z / det);
  v3 ip = vneg(vadd(vscale(p.x, ix), vadd(vscale(p.y, iy), vscale(p.z, iz))));
  v3 lo = vfrom(S->vb_lo), hi = vfrom(S->vb_hi);               /* OptixRenderer.cpp:330-332 */
  *wlo = vadd(vscale(lo.x, ix), vadd(vscale(lo.y, iy), vadd(vscale(lo.z, iz), ip)));
  *whi = vadd(vscale(hi.x, ix), vadd(vscale(hi.y, iy), vadd(vscale(hi.z, iz), ip)));
}

/* exabrick.cu:1345-1406 traceContourRay */
static SurfacePRD trace_contour_ray(Ctx *C, Ray ray, v3 normal, float offset, int channel)
{
  SurfacePRD prd; memset(&prd, 0, sizeof(prd));
  prd.primID = -1;      /* left uninitialised by the reference when the plane is missed */
  prd.t_hit = ray.tmax;
  v3 pts[6];
  int isectCnt;
  intersect_box_plane(V3(0.f, 0.f, 0.f), V3(1.f, 1.f, 1.f), normal, offset, pts, &isectCnt);
  v3 wlo, whi;
  world_space_bounds(C->S, C->fs, &wlo, &whi);
  for (int i = 0; i < isectCnt; ++i) {                          /* scale to world bounds :1359-1362 */
    pts[i] = vmul(pts[i], vsub(whi, wlo));
    pts[i] = vadd(pts[i], wlo);
  }
  float t = -1.f;
  for (int i = 0; i < isectCnt - 1; ++i) {                      /* cyclical selection sort :1367-1382 */
    int minIdx = i;
    for (int j = i + 1; j < isectCnt; ++j) {
      v3 v = vcross(vsub(pts[j], pts[0]), vsub(pts[minIdx], pts[0]));
      if (vdot(v, normal) < 0.f) minIdx = j;
    }
    v3 tmp = pts[i]; pts[i] = pts[minIdx]; pts[minIdx] = tmp;
  }
  for (int i = 2; i < isectCnt; ++i) {                          /* fan :1384-1391 */
    v3 v1 = pts[0], e1 = vsub(pts[i - 1], v1), e2 = vsub(pts[i], v1);
    float tt = intersect_ray_triangle(&ray, v1, e1, e2);
    if (tt >= 0.f && (tt < t || t < 0.f)) t = tt;
  }
  if (t < 0.f) return prd;
  float value = sample_point_with_inf_ray(C, vadd(ray.org, vscale(t, ray.dir)), 0);   /* :1396, channel 0 */
  v4 sample = lookup_xf(C->S, C->fs, value, channel);
  prd.primID = PRIMID_PLANE;
  prd.t_hit = t;
  prd.Ng = normal;
  prd.ambient = 0.f;
  prd.baseColor = V3(sample.x, sample.y, sample.z);
  return prd;
}

/* exabrick.cu:1475-1529 traceSurfaces: contour planes and implicit iso-surfaces (meshes and
 * streamlines are SURVEY 8f rank 4) */
/* ------------------------------------------------------------------ */
/* streamlines                                      */
/* ------------------------------------------------------------------ */
#define PRIMID_STREAMLINE (-25)

/* exabrick.cu:440-502 intersectRoundedCone */
static int intersect_rounded_cone(v3 pa, v3 pb, float ra, float rb, const Ray *ray, float *hit_t, v3 *isec_normal)
{
  v3 ro = ray->org;
  const v3 rd = ray->dir;
  float minDist = fmaxf(0.f, fminf(vlength(vsub(pa, ro)) - ra, vlength(vsub(pb, ro)) - rb));
  ro = vadd(ro, vscale(minDist, rd));
  v3 ba = vsub(pb, pa), oa = vsub(ro, pa), ob = vsub(ro, pb);
  float rr = ra - rb;
  float m0 = vdot(ba, ba), m1 = vdot(ba, oa), m2 = vdot(ba, rd), m3 = vdot(rd, oa), m5 = vdot(oa, oa);
  float m6 = vdot(ob, rd), m7 = vdot(ob, ob);
  (void)m6; (void)m7;
  float d2 = m0 - rr * rr;
  float k2 = d2 - m2 * m2;
  float k1 = d2 * m3 - m1 * m2 + m2 * rr * ra;
  float k0 = d2 * m5 - m1 * m1 + m1 * rr * ra * 2.0f - m0 * ra * ra;
  float h = k1 * k1 - k0 * k2;
  if (h < 0.0f) return 0;
  float t = (-sqrtf(h) - k1) / k2;
  float y = m1 - ra * rr + t * m2;
  if (y > 0.0f && y < d2) {
    *hit_t = minDist + t;
    *isec_normal = vsub(vscale(d2, vadd(oa, vscale(t, rd))), vscale(y, ba));
    return 1;
  }
  float h1 = m3 * m3 - m5 + ra * ra;                          /* caps */
  if (h1 > 0.0f) {
    t = -m3 - sqrtf(h1);
    *hit_t = minDist + t;
    *isec_normal = vdiv(vadd(oa, vscale(t, rd)), v3s(ra));
    return 1;
  }
  return 0;
}

/* the streamlineBVH trace of traceSurfaces (:1503-1512) with the Streamline bounds/intersect programs
 * (:504-570): segments hidden by the bounds program are skipped, closest accepted t in [tmin, best],
 * lowest primitive id on a tie.  Brute force. */
static void trace_streamlines(Ctx *C, const Ray *ray, SurfacePRD *prd)
{
  const OrScene *S = C->S;
  if (!S->traces) return;
  const int NT = S->numTimesteps, t = S->timestep;
  const long nprims = (long)S->numTraces * (NT - 1);
  float best = 2e10f;                                           /* streamlinePRD.t_hit = 2e10f (:1507) */
  long hit = -1;
  v3 bestN = v3s(0.f);
  Ray r = *ray;
  for (long p = 0; p < nprims; p++) {
    if ((int)(p % NT) >= t - 1) continue;                       /* :546-551 */
    const v3 pa = vfrom(&S->traces[3 * p]), pb = vfrom(&S->traces[3 * (p + 1)]);
    if (!(pa.x < 2e10f && pb.x < 2e10f)) continue;              /* :559-570 */
    float th; v3 n;
    if (!intersect_rounded_cone(pa, pb, 2.f, 2.f, &r, &th, &n)) continue;
    if (th >= ray->tmin && th <= ray->tmax && th < best) { best = th; hit = p; bestN = n; }
  }
  if (hit >= 0 && best < prd->t_hit) {                          /* :1510-1511 */
    prd->primID = PRIMID_STREAMLINE;
    prd->t_hit = best;
    prd->Ng = vnormalize(bestN);
    prd->baseColor = v3s(.8f);
    prd->ambient = 0.f;   /* never written by the reference for streamlines (uninitialised there) */
  }
}

/* exabrick.cu:945-963 sampleDirection */
static int sample_direction(Ctx *C, v3 pos, v3 *result)
{
  Ray ray = { pos, V3(1.f, 1.f, 1.f), 0.f, 2e-10f };
  RegionHit prd = trace_region(C->S, C->volActive, &ray);
  if (prd.leafID < 0) return 0;                                 /* reference: region[-1], undefined */
  float r[3] = {0.f, 0.f, 0.f};
  for (int i = 0; i < 3; ++i)
    if (!sample_point(C, &r[i], prd.leafID, pos, C->S->tracerChannels[i])) { *result = V3(r[0], r[1], r[2]); return 0; }
  *result = V3(r[0], r[1], r[2]);
  return 1;
}

/* exabrick.cu:1531-1574 computeTraces for trace i (the thread with pixelIdx == i) */
static void compute_trace(Ctx *C, int i)
{
  OrScene *S = (OrScene *)C->S;
  const int t = S->timestep, NT = S->numTimesteps;
  if (!(t >= 1 && t < NT && i < S->numTraces && S->traces)) return;   /* t == 0 would read traces[-1] (undefined in the reference) */
  v3 wlo, whi;
  world_space_bounds(S, C->fs, &wlo, &whi);
  v3 p = vfrom(&S->traces[3 * ((size_t)i * NT + (t - 1))]);
  const v3 pp = p;
  if (p.x < 2e10f) {
    int valid = 1;
    v3 k1 = v3s(0.f), k2 = v3s(0.f), k3 = v3s(0.f), k4 = v3s(0.f);
    valid &= sample_direction(C, p, &k1);
    k1 = vscale(S->steplen, k1);
    v3 ptry1 = vadd(p, vscale(.5f, k1));
    valid &= sample_direction(C, ptry1, &k2);
    k2 = vscale(S->steplen, k2);
    v3 ptry2 = vadd(p, vscale(.5f, k2));
    valid &= sample_direction(C, ptry2, &k3);
    k3 = vscale(S->steplen, k3);
    v3 ptry3 = vadd(p, k3);
    valid &= sample_direction(C, ptry3, &k4);
    k4 = vscale(S->steplen, k4);
    p = vadd(p, vscale(1 / 6.f, vadd(vadd(vadd(k1, vscale(2.f, k2)), vscale(2.f, k3)), k4)));
    const int inside = p.x >= wlo.x && p.y >= wlo.y && p.z >= wlo.z && p.x <= whi.x && p.y <= whi.y && p.z <= whi.z;
    if (!valid || !inside || vlength(vsub(p, pp)) < 1e-10f) p = v3s(2e10f);
  }
  float *dst = &S->traces[3 * ((size_t)i * NT + t)];
  dst[0] = p.x; dst[1] = p.y; dst[2] = p.z;
}

/* The triangle trace of traceSurfaces (:1483-1486) and its closest-hit program (:420-433).  OptiX's
 * built-in triangle intersector is not observable; the oracle uses the reference's own
 * intersectRayTriangle (:1316-1343, Moeller-Trumbore) with t in (tmin,tmax), closest t, lowest
 * triangle index on a tie.  Brute force over all triangles. */
static void trace_meshes(Ctx *C, const Ray *ray, SurfacePRD *prd)
{
  const OrScene *S = C->S;
  float best = ray->tmax;
  long hit = -1;
  for (size_t i = 0; i < S->numTris; i++) {
    const int32_t *t = &S->meshTris[3 * i];
    const v3 A = vfrom(&S->meshVerts[3 * t[0]]), B = vfrom(&S->meshVerts[3 * t[1]]), Cc = vfrom(&S->meshVerts[3 * t[2]]);
    const float tt = intersect_ray_triangle(ray, A, vsub(B, A), vsub(Cc, A));
    if (tt > ray->tmin && tt < best) { best = tt; hit = (long)i; }
  }
  if (hit >= 0) {
    const int32_t *t = &S->meshTris[3 * hit];
    const v3 A = vfrom(&S->meshVerts[3 * t[0]]), B = vfrom(&S->meshVerts[3 * t[1]]), Cc = vfrom(&S->meshVerts[3 * t[2]]);
    prd->t_hit = best;
    prd->primID = (int)hit;
    prd->Ng = vnormalize(vcross(vsub(B, A), vsub(Cc, A)));
    prd->ambient = .2f;
    prd->baseColor = v3s(.8f);
  }
}

static void trace_surfaces(Ctx *C, Ray ray, SurfacePRD *prd, int withContourPlanes)
{
  prd->primID = -1;
  prd->t_hit = ray.tmax;
  if (C->S->numTris) trace_meshes(C, &ray, prd);                 /* ST_MESHES, also for AO rays */
  if (withContourPlanes) {
    for (int i = 0; i < OR_MAX_CONTOUR_PLANES; ++i) {
      if (C->fs->contour[i].enabled) {
        SurfacePRD contourPRD = trace_contour_ray(C, ray, vfrom(C->fs->contour[i].normal),
                                                  C->fs->contour[i].offset, C->fs->contour[i].channel);
        if (contourPRD.primID == PRIMID_PLANE && contourPRD.t_hit < prd->t_hit) *prd = contourPRD;
      }
    }
  }
  trace_streamlines(C, &ray, prd);                              /* ST_STREAMLINES :1503-1512 */
  int activeIsoSurfaces = 0;
  for (int i = 0; i < OR_MAX_ISO_SURFACES; i++) activeIsoSurfaces |= C->fs->iso[i].enabled;
  if (activeIsoSurfaces) {
    SurfacePRD isoPRD = trace_iso_ray(C, ray, 0.f);
    if (isoPRD.primID == PRIMID_ISOSURFACE && isoPRD.t_hit < prd->t_hit) *prd = isoPRD;
  }
}

/* exabrick.cu:78-94 */
static void make_orthonormal_basis(v3 *u, v3 *v, v3 w)
{
  *v = fabsf(w.x) > fabsf(w.y) ? vnormalize(V3(-w.z, 0.f, w.x)) : vnormalize(V3(0.f, w.z, -w.y));
  *u = vcross(*v, w);
}
static v3 cosine_sample_hemisphere(float u1, float u2)
{
  float r = sqrtf(u1);
  float theta = 2.f * (float)M_PI * u2;
  return V3(r * cosf(theta), r * sinf(theta), sqrtf(1.f - u1));
}

/* exabrick.cu:1576-1720 renderFrame for one pixel */
static void render_pixel(Ctx *C, int px, int py, int W, int H, uint32_t *rgba, float *accum4)
{
  const OrFrameState *fs = C->fs;
  const int pixelIdx = px + W * py;
  const int frameID = fs->frameID;
  Lcg rnd;
  lcg_init(&rnd, (uint32_t)(frameID * W * H) + (uint32_t)px, (uint32_t)py);   /* :1591-1592 */
  const float sx = (float)px + lcg_next(&rnd);                               /* :1594, x then y */
  const float sy = (float)py + lcg_next(&rnd);
  Ray ray;                                                                   /* Camera.h:27-44 */
  ray.org = vfrom(fs->cam_pos);
  ray.dir = vnormalize(vadd(vadd(vfrom(fs->cam_dir00), vscale(sx, vfrom(fs->cam_dirDu))),
                            vscale(sy, vfrom(fs->cam_dirDv))));
  ray.tmin = 1e-6f; ray.tmax = 1e8f;

  SurfacePRD surface; memset(&surface, 0, sizeof(surface));
  trace_surfaces(C, ray, &surface, 1);                                       /* :1601 ST_ALL_SURFACES */

  v3 bgColor = v3s(0.f);
  if (surface.primID >= 0 || surface.primID == PRIMID_ISOSURFACE || surface.primID == PRIMID_PLANE
      || surface.primID == PRIMID_STREAMLINE) {                                   /* :1604 */
    const int shade = surface.primID >= 0 || surface.primID == PRIMID_PLANE || surface.primID == PRIMID_STREAMLINE
                   || (surface.primID == PRIMID_ISOSURFACE && C->P->gradientShadingISO);
    if (shade && vlength(surface.Ng) > 0.f) {
      const float AO_Radius = fs->ao.length;
      const int AO_Samples = fs->ao.enabled ? 2 : 0;
      v3 isect_pos = vadd(ray.org, vscale(surface.t_hit, ray.dir));
      v3 u, v, w = surface.Ng;
      make_orthonormal_basis(&u, &v, w);
      int hitCnt = 0;
      for (int i = 0; i < AO_Samples; ++i) {
        float r1 = lcg_next(&rnd), r2 = lcg_next(&rnd);                      /* :1624 */
        v3 sp = cosine_sample_hemisphere(r1, r2);
        v3 dir = vnormalize(vadd(vadd(vscale(sp.x, u), vscale(sp.y, v)), vscale(sp.z, w)));
        Ray ao_ray = {isect_pos, dir, 1e-4f, AO_Radius};
        SurfacePRD ao;
        trace_surfaces(C, ao_ray, &ao, 0);                                   /* :1637-1639 no contour planes */
        if (ao.primID >= 0 || ao.primID == PRIMID_ISOSURFACE || ao.primID == PRIMID_PLANE
            || ao.primID == PRIMID_STREAMLINE) hitCnt++;
      }
      float shadow = fs->ao.enabled ? (float)hitCnt / AO_Samples : 0.f;
      /* :1646-1648  ambient + baseColor*fabs(dot(dir,Ng))*(1-shadow), left to right */
      const float fd = fabsf(vdot(ray.dir, surface.Ng));
      const float ns = 1.f - shadow;
      bgColor = V3(surface.ambient + surface.baseColor.x * fd * ns,
                   surface.ambient + surface.baseColor.y * fd * ns,
                   surface.ambient + surface.baseColor.z * fd * ns);
    } else {
      bgColor = surface.baseColor;
    }
  }

  v4 pixelColor = {0.f, 0.f, 0.f, 0.f};
  float interleavedSamplingOffset = lcg_next(&rnd);                          /* :1655 */

  ray.tmax = surface.t_hit;                                                  /* :1657 */
  if (fs->clipBox.enabled) {                                                 /* clipRay :1258-1265 */
    float a, b;
    box_test(&ray, fs->clipBox.lo, fs->clipBox.hi, &a, &b);
    ray.tmin = a; ray.tmax = b;
  }
  surface.t_hit = ray.tmax;

  ray.org = xfm_point(fs, ray.org);                                          /* :1664-1668 */
  ray.dir = xfm_vector(fs, ray.dir);
  const float dt_scale = vlength(ray.dir);
  ray.dir = vnormalize(ray.dir);

  float alreadyIntegratedDistance = dt_scale * ray.tmin;
  for (;;) {                                                                 /* :1675-1699 */
    ray.tmin = alreadyIntegratedDistance;
    ray.tmax = surface.t_hit * dt_scale;
    RegionHit prd = trace_region(C->S, C->volActive, &ray);
    if (prd.leafID < 0) break;
    C->st.segments++;
    IntegrationResult ir = {&pixelColor, -1.f, {0.f, 0.f, 0.f}};
    integrate_brick(C, C->P->gradientShadingDVR != 0, &ir, interleavedSamplingOffset, &ray,
                    prd.leafID, prd.t0, prd.t1, C->P->numPrimaryChannels);
    if (pixelColor.w >= TERMINATION_THRESHOLD) {
      pixelColor.x = pixelColor.x * pixelColor.w;                            /* :1695 */
      pixelColor.y = pixelColor.y * pixelColor.w;
      pixelColor.z = pixelColor.z * pixelColor.w;
      pixelColor.w = 1.f;
      break;
    }
    alreadyIntegratedDistance = prd.t1 * (1.0000001f);
  }

  v3 color = V3(pixelColor.w * pixelColor.x + (1.f - pixelColor.w) * bgColor.x,   /* :1701 */
                pixelColor.w * pixelColor.y + (1.f - pixelColor.w) * bgColor.y,
                pixelColor.w * pixelColor.z + (1.f - pixelColor.w) * bgColor.z);
  /* clockScale heat-map (:1703-1707) reads the GPU cycle counter; not reproducible, not restated here (the HIP
   * kernels implement it; tests/test_gpu_parity.py::test_clock_heat_map checks its range and that g, b are untouched) */
  if (frameID > 0) {                                                         /* :1709-1710 */
    color.x += accum4[4 * pixelIdx + 0];
    color.y += accum4[4 * pixelIdx + 1];
    color.z += accum4[4 * pixelIdx + 2];
  }
  accum4[4 * pixelIdx + 0] = color.x;                                        /* :1712 */
  accum4[4 * pixelIdx + 1] = color.y;
  accum4[4 * pixelIdx + 2] = color.z;
  accum4[4 * pixelIdx + 3] = 1.f;
  const float div = frameID + 1.f;                                           /* :1714 */
  color = V3(color.x / div, color.y / div, color.z / div);
  rgba[pixelIdx] = or_make_rgba8(or_linear_to_srgb(color.x), or_linear_to_srgb(color.y),
                                 or_linear_to_srgb(color.z));
}

/* ------------------------------------------------------------------ */
/* frame driver: rows handed out to pthreads                           */
/* ------------------------------------------------------------------ */
typedef struct {
  Ctx C; int W, H, x0, y0, x1, y1; uint32_t *rgba; float *accum4;
  volatile int *nextRow; pthread_mutex_t *mu;
} Job;

static void *worker(void *arg)
{
  /* work items are 8x8 pixel blocks handed out from a shared counter */
  Job *J = (Job *)arg;
  const int bw = (J->x1 - J->x0 + 7) / 8, bh = (J->y1 - J->y0 + 7) / 8;
  for (;;) {
    pthread_mutex_lock(J->mu);
    int b = (*J->nextRow)++;
    pthread_mutex_unlock(J->mu);
    if (b >= bw * bh) break;
    const int bx0 = J->x0 + (b % bw) * 8, by0 = J->y0 + (b / bw) * 8;
    for (int y = by0; y < by0 + 8 && y < J->y1; y++)
      for (int x = bx0; x < bx0 + 8 && x < J->x1; x++) render_pixel(&J->C, x, y, J->W, J->H, J->rgba, J->accum4);
  }
  return NULL;
}

void or_render(const OrScene *S, const OrFrameState *fs, const OrParams *P,
               int W, int H, int x0, int y0, int x1, int y1,
               uint32_t *rgba, float *accum4, OrStats *stats, int nthreads)
{
  if (nthreads <= 0) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  uint8_t *volActive = (uint8_t *)xmalloc(S->numRegions), *isoActive = (uint8_t *)xmalloc(S->numRegions);
  or_volume_active(S, fs, P, volActive);
  or_iso_active(S, fs, isoActive);
  if (S->tracerEnabled && S->traces) {          /* renderFrame :1581-1582; trace i is advanced by the thread of pixel i */
    Ctx tc; memset(&tc, 0, sizeof(tc));
    tc.S = S; tc.fs = fs; tc.P = P; tc.volActive = volActive; tc.isoActive = isoActive;
    for (int i = 0; i < S->numTraces && i < W * H; i++) compute_trace(&tc, i);
  }
  pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
  volatile int nextRow = 0;
  Job *jobs = (Job *)xmalloc((size_t)nthreads * sizeof(Job));
  pthread_t *th = (pthread_t *)xmalloc((size_t)nthreads * sizeof(pthread_t));
  for (int i = 0; i < nthreads; i++) {
    memset(&jobs[i], 0, sizeof(Job));
    jobs[i].C.S = S; jobs[i].C.fs = fs; jobs[i].C.P = P;
    jobs[i].C.volActive = volActive; jobs[i].C.isoActive = isoActive;
    jobs[i].W = W; jobs[i].H = H; jobs[i].x0 = x0; jobs[i].y0 = y0; jobs[i].x1 = x1; jobs[i].y1 = y1;
    jobs[i].rgba = rgba; jobs[i].accum4 = accum4; jobs[i].nextRow = &nextRow; jobs[i].mu = &mu;
  }
  if (nthreads == 1) worker(&jobs[0]);
  else {
    for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, worker, &jobs[i]);
    for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
  }
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    for (int i = 0; i < nthreads; i++) {
      stats->segments += jobs[i].C.st.segments;
      stats->sample_evals += jobs[i].C.st.sample_evals;
      stats->samples += jobs[i].C.st.samples;
      stats->brick_visits += jobs[i].C.st.brick_visits;
      stats->corner_loads += jobs[i].C.st.corner_loads;
      stats->iso_segments += jobs[i].C.st.iso_segments;
      stats->iso_evals += jobs[i].C.st.iso_evals;
    }
  }
  free(jobs); free(th); free(volActive); free(isoActive);
}
