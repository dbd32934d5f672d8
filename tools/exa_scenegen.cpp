// exa_scenegen.cpp — seeded procedural AMR scenes emitted directly in ExaBricks
// form (bench/test input generator; not part of the render path).
//
// None of the reference's data sets (exajet, landing gear, LANL deep-water) exist
// in the build container or on the GPU box, so the benchmark configurations are
// deterministic stand-ins: an octree of B^3-cell blocks refined
// toward a signed-distance feature; every leaf block is one brick (dense, single
// level, disjoint, power-of-two cell width) exactly as `exaBuilder` would emit.
//
//   kind 0  "lanl-like":        expanding spherical shell + column (impact plume)
//   kind 1  "landing-gear-like": struts / cylinders
//   kind 2  "exajet-like":      fuselage ellipsoid + swept wings + tail + wake cone
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

namespace {

struct Params {
  uint64_t seed;
  int32_t  rootN[3];   // root blocks per axis
  int32_t  B;          // cells per block edge
  int32_t  levels;     // root blocks are at level levels-1
  int32_t  kind;
  float    band;       // refinement band scale (bigger = more fine cells)
  int32_t  numFields;
  int32_t  threads;
};

inline uint64_t splitmix64(uint64_t x)
{
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline float hash01(int64_t ix, int64_t iy, int64_t iz, uint64_t seed)
{
  const uint64_t k = uint64_t(ix) * 0x9E3779B1ull ^ uint64_t(iy) * 0x85EBCA77ull ^ uint64_t(iz) * 0xC2B2AE3Dull ^ seed;
  return float(splitmix64(k) >> 40) * (1.f / 16777216.f);
}
inline float valueNoise(float x, float y, float z, float invCell, uint64_t seed)
{
  x *= invCell; y *= invCell; z *= invCell;
  const float fx0 = std::floor(x), fy0 = std::floor(y), fz0 = std::floor(z);
  float fx = x - fx0, fy = y - fy0, fz = z - fz0;
  fx = fx * fx * (3.f - 2.f * fx); fy = fy * fy * (3.f - 2.f * fy); fz = fz * fz * (3.f - 2.f * fz);
  const int64_t ix = (int64_t)fx0, iy = (int64_t)fy0, iz = (int64_t)fz0;
  float c[2][2][2];
  for (int dz = 0; dz < 2; dz++) for (int dy = 0; dy < 2; dy++) for (int dx = 0; dx < 2; dx++)
    c[dz][dy][dx] = hash01(ix + dx, iy + dy, iz + dz, seed);
  const float x00 = c[0][0][0] + fx * (c[0][0][1] - c[0][0][0]), x10 = c[0][1][0] + fx * (c[0][1][1] - c[0][1][0]);
  const float x01 = c[1][0][0] + fx * (c[1][0][1] - c[1][0][0]), x11 = c[1][1][0] + fx * (c[1][1][1] - c[1][1][0]);
  const float y0 = x00 + fy * (x10 - x00), y1 = x01 + fy * (x11 - x01);
  return y0 + fz * (y1 - y0);
}

struct Feature {
  int kind;
  float ext[3];     // domain extent in finest cells
  float unit;       // min extent

  static float sdBox(float px, float py, float pz, float hx, float hy, float hz)
  {
    const float qx = std::fabs(px) - hx, qy = std::fabs(py) - hy, qz = std::fabs(pz) - hz;
    const float ox = std::max(qx, 0.f), oy = std::max(qy, 0.f), oz = std::max(qz, 0.f);
    return std::sqrt(ox * ox + oy * oy + oz * oz) + std::min(std::max(qx, std::max(qy, qz)), 0.f);
  }
  static float sdCyl(float px, float py, float pz, float r, float h)   // axis = y
  {
    const float dx = std::sqrt(px * px + pz * pz) - r, dy = std::fabs(py) - h;
    const float ox = std::max(dx, 0.f), oy = std::max(dy, 0.f);
    return std::min(std::max(dx, dy), 0.f) + std::sqrt(ox * ox + oy * oy);
  }
  // signed distance (finest-cell units) to the solid feature; wake returns a second distance
  float body(float x, float y, float z, float &wake) const
  {
    const float u = unit;
    wake = 1e30f;
    if (kind == 0) {
      const float cx = 0.45f * ext[0], cy = 0.5f * ext[1], cz = 0.5f * ext[2];
      const float dx = x - cx, dy = y - cy, dz = z - cz;
      const float shell = std::fabs(std::sqrt(dx * dx + dy * dy + dz * dz) - 0.33f * u) - 0.01f * u;
      const float col = std::sqrt(dx * dx + dz * dz) - 0.05f * u;
      wake = col;
      return shell;
    }
    if (kind == 1) {
      const float cx = 0.5f * ext[0], cy = 0.5f * ext[1], cz = 0.5f * ext[2];
      float d = sdCyl(x - cx, y - cy, z - cz, 0.035f * u, 0.38f * u);                       // main strut
      d = std::min(d, sdCyl(y - (cy - 0.3f * u), x - cx, z - cz, 0.11f * u, 0.05f * u));      // wheel (axis x)
      d = std::min(d, sdCyl(x - (cx + 0.12f * u), (y - cy) * 0.8f + (z - cz) * 0.6f, (z - cz) * 0.8f - (y - cy) * 0.6f,
                            0.02f * u, 0.3f * u));                                            // brace
      d = std::min(d, sdBox(x - cx, y - (cy + 0.36f * u), z - cz, 0.2f * u, 0.02f * u, 0.12f * u)); // door
      return d;
    }
    // exajet-like, flying toward -x
    const float cx = 0.38f * ext[0], cy = 0.5f * ext[1], cz = 0.5f * ext[2];
    const float dx = x - cx, dy = y - cy, dz = z - cz;
    const float a = 0.26f * ext[0], b = 0.035f * u * 2.f, c = 0.035f * u * 2.f;
    const float k = std::sqrt((dx / a) * (dx / a) + (dy / b) * (dy / b) + (dz / c) * (dz / c));
    float d = (k - 1.f) * std::min(b, c);                                                     // fuselage (approx. sdf)
    const float sw = 0.5f;                                                                    // wing sweep
    const float az = std::fabs(dz);
    const float wx = (dx - 0.02f * ext[0]) - sw * az;
    d = std::min(d, sdBox(wx, dy + 0.01f * u, az - 0.22f * u, 0.035f * ext[0], 0.008f * u, 0.22f * u)); // wings
    const float tx = (dx - 0.21f * ext[0]) - sw * az;
    d = std::min(d, sdBox(tx, dy - 0.02f * u, az - 0.08f * u, 0.018f * ext[0], 0.006f * u, 0.08f * u)); // tailplane
    d = std::min(d, sdBox((dx - 0.21f * ext[0]) - 0.6f * (dy - 0.06f * u), dy - 0.09f * u, dz,
                          0.02f * ext[0], 0.07f * u, 0.005f * u));                            // fin
    // wake cone behind the wings
    const float wxs = dx - 0.05f * ext[0];
    if (wxs > 0.f) {
      const float r = 0.05f * u + 0.10f * wxs;
      wake = std::sqrt(dy * dy + dz * dz) - r;
    }
    return d;
  }
};

struct Gen {
  Params P;
  Feature F;
  std::vector<int32_t> bricks7;
  std::vector<std::unique_ptr<float[]>> fields;   // uninitialised: first touched by the fill threads
  std::unique_ptr<int32_t[]> cellIDs;
  uint64_t numCells = 0;

  bool wantRefine(int x, int y, int z, int l) const
  {
    const float e = float(P.B << l);
    float wake;
    const float d = F.body(x + 0.5f * e, y + 0.5f * e, z + 0.5f * e, wake);
    const float half = 0.87f * e;                           // ~ half block diagonal
    // nested bands: level l -> l-1 happens within band(l) of the surface
    const float band = P.band * F.unit * (0.006f * float(1 << (2 * (l - 1))) + 0.004f);
    if (std::fabs(d) < band + half) return true;
    if (l >= 2 && wake < 0.5f * half) return true;          // wake interior refined down to level 1
    return false;
  }

  void run()
  {
    const int top = P.levels - 1;
    F.kind = P.kind;
    for (int k = 0; k < 3; k++) F.ext[k] = float(P.rootN[k] * (P.B << top));
    F.unit = std::min(F.ext[0], std::min(F.ext[1], F.ext[2]));
    // octree walk, depth first in Morton order (x fastest), explicit stack
    struct Blk { int x, y, z, l; };
    std::vector<Blk> stack;
    for (int z = P.rootN[2] - 1; z >= 0; z--)
      for (int y = P.rootN[1] - 1; y >= 0; y--)
        for (int x = P.rootN[0] - 1; x >= 0; x--) stack.push_back({ x * (P.B << top), y * (P.B << top), z * (P.B << top), top });
    while (!stack.empty()) {
      const Blk b = stack.back();
      stack.pop_back();
      if (b.l > 0 && wantRefine(b.x, b.y, b.z, b.l)) {
        const int h = (P.B << b.l) / 2;
        for (int c = 7; c >= 0; c--) stack.push_back({ b.x + (c & 1) * h, b.y + ((c >> 1) & 1) * h, b.z + (c >> 2) * h, b.l - 1 });
      } else {
        const int32_t rec[7] = { P.B, P.B, P.B, b.x, b.y, b.z, b.l };
        bricks7.insert(bricks7.end(), rec, rec + 7);
      }
    }
    const size_t nb = bricks7.size() / 7;
    const size_t per = size_t(P.B) * P.B * P.B;
    numCells = nb * per;
  }

  // bricks given from outside (e.g. the bricks exaBuilder made of this scene's cells): same domain, same field functions
  void adopt(const int32_t *b7, size_t nb)
  {
    const int top = P.levels - 1;
    F.kind = P.kind;
    for (int k = 0; k < 3; k++) F.ext[k] = float(P.rootN[k] * (P.B << top));
    F.unit = std::min(F.ext[0], std::min(F.ext[1], F.ext[2]));
    bricks7.assign(b7, b7 + 7 * nb);
    numCells = 0;
    for (size_t b = 0; b < nb; b++) numCells += uint64_t(b7[7 * b]) * uint64_t(b7[7 * b + 1]) * uint64_t(b7[7 * b + 2]);
  }

  void fill()
  {
    const size_t nb = bricks7.size() / 7;
    std::vector<size_t> begin(nb + 1, 0);
    for (size_t b = 0; b < nb; b++) begin[b + 1] = begin[b] + size_t(bricks7[7 * b]) * size_t(bricks7[7 * b + 1]) * size_t(bricks7[7 * b + 2]);
    fields.clear();
    for (int f = 0; f < P.numFields; f++) fields.emplace_back(new float[numCells]);
    cellIDs.reset(new int32_t[numCells]);
    const int nt = std::max(1, P.threads);
    std::atomic<size_t> next{0};
    auto work = [&]() {
      for (;;) {
        const size_t b0 = next.fetch_add(64);
        if (b0 >= nb) break;
        const size_t b1 = std::min(nb, b0 + 64);
        for (size_t b = b0; b < b1; b++) {
          const int32_t *r = &bricks7[7 * b];
          const float cw = float(1 << r[6]);
          size_t i = begin[b];
          for (int iz = 0; iz < r[2]; iz++)
            for (int iy = 0; iy < r[1]; iy++)
              for (int ix = 0; ix < r[0]; ix++, i++) {
                const float x = r[3] + (ix + 0.5f) * cw, y = r[4] + (iy + 0.5f) * cw, z = r[5] + (iz + 0.5f) * cw;
                float wake;
                const float d = F.body(x, y, z, wake);
                // "vorticity-like": strong in a thin layer around the body, turbulent in
                // the wake, exactly 0 in the free stream and inside the body
                const float layer = (d < -0.004f * F.unit || d > 0.12f * F.unit)
                                        ? 0.f : std::exp(-std::max(d, 0.f) / (0.012f * F.unit));
                const bool inWake = wake < 0.02f * F.unit;
                float s0 = 0.f, n1 = 0.f;
                if (layer > 0.f || inWake || P.numFields > 1) {
                  const float n2 = valueNoise(x, y, z, 1.f / (0.021f * F.unit), P.seed + 23);
                  float wk = 0.f;
                  if (inWake || P.numFields > 1) n1 = valueNoise(x, y, z, 1.f / (0.09f * F.unit), P.seed + 11);
                  if (inWake) {
                    const float in = std::min(1.f, (0.02f * F.unit - wake) / (0.04f * F.unit));
                    wk = in * (0.25f + 0.75f * n1) * (0.4f + 0.6f * n2);
                  }
                  s0 = std::min(1.f, 0.85f * layer * (0.6f + 0.4f * n2) + 0.7f * wk);
                }
                fields[0][i] = s0;
                // further fields: a second quantity living in the same flow features
                for (int f = 1; f < P.numFields; f++)
                  fields[f][i] = std::min(1.f, layer + (inWake ? 0.6f : 0.f))
                               * (0.5f * n1 + 0.5f * valueNoise(x, y, z, 1.f / (0.05f * F.unit), P.seed + 31 * f));
                cellIDs[i] = (int32_t)i;
              }
        }
      }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++) pool.emplace_back(work);
    for (auto &t : pool) t.join();
  }
};

} // namespace

extern "C" {

struct ExaSceneGen { Gen g; };

// count-only when fill == 0 (fast: tune parameters to a target cell count)
int exa_scenegen_create(uint64_t seed, const int32_t rootN[3], int32_t B, int32_t levels, int32_t kind,
                        float band, int32_t numFields, int32_t threads, int32_t fill, ExaSceneGen **out)
{
  if (!out || B < 1 || levels < 1 || levels > 8 || numFields < 1 || numFields > 10) return 1;
  ExaSceneGen *G = new ExaSceneGen;
  G->g.P = Params{ seed, { rootN[0], rootN[1], rootN[2] }, B, levels, kind, band, numFields,
                   threads > 0 ? threads : (int)std::max(1u, std::thread::hardware_concurrency()) };
  G->g.run();
  if (G->g.numCells > 0x7fffffffull) { if (fill) { delete G; return 2; } }   // 32-bit brick offsets (Brick.h:70)
  if (fill) G->g.fill();
  *out = G;
  return 0;
}
// the same scene (domain, feature, field functions of `kind` / `seed` / `rootN` / `B` / `levels`) on a brick list given from
// outside, bricks7 = numBricks x {size.xyz, lower.xyz, level}: e.g. the bricks exaBuilder made of the scene's cells
int exa_scenegen_create_from_bricks(uint64_t seed, const int32_t rootN[3], int32_t B, int32_t levels, int32_t kind,
                                    float band, int32_t numFields, int32_t threads, const int32_t *bricks7, uint64_t numBricks,
                                    ExaSceneGen **out)
{
  if (!out || !bricks7 || numBricks == 0 || B < 1 || levels < 1 || levels > 8 || numFields < 1 || numFields > 10) return 1;
  for (uint64_t b = 0; b < numBricks; b++) {
    const int32_t *r = bricks7 + 7 * b;
    if (r[0] < 1 || r[1] < 1 || r[2] < 1 || r[6] < 0 || r[6] > 30) return 1;
  }
  ExaSceneGen *G = new ExaSceneGen;
  G->g.P = Params{ seed, { rootN[0], rootN[1], rootN[2] }, B, levels, kind, band, numFields,
                   threads > 0 ? threads : (int)std::max(1u, std::thread::hardware_concurrency()) };
  G->g.adopt(bricks7, (size_t)numBricks);
  if (G->g.numCells > 0x7fffffffull) { delete G; return 2; }
  G->g.fill();
  *out = G;
  return 0;
}
void exa_scenegen_destroy(ExaSceneGen *G) { delete G; }
uint64_t exa_scenegen_num_bricks(const ExaSceneGen *G) { return G->g.bricks7.size() / 7; }
uint64_t exa_scenegen_num_cells(const ExaSceneGen *G) { return G->g.numCells; }
const int32_t *exa_scenegen_bricks7(const ExaSceneGen *G) { return G->g.bricks7.data(); }
const int32_t *exa_scenegen_cell_ids(const ExaSceneGen *G) { return G->g.cellIDs.get(); }
const float *exa_scenegen_field(const ExaSceneGen *G, int32_t f) { return G->g.fields[f].get(); }
void exa_scenegen_level_histogram(const ExaSceneGen *G, uint64_t hist[8])
{
  for (int i = 0; i < 8; i++) hist[i] = 0;
  for (size_t b = 0; b < G->g.bricks7.size() / 7; b++) hist[G->g.bricks7[7 * b + 6] & 7]++;
}

} // extern "C"
