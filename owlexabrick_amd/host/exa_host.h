// exa_host.h — C++ host layer above the C ABI (include/exa_hip.h).
//
// Mirrors the interface the reference's viewer programs against, so a host written
// for `exa::OptixRenderer` recompiles against `exa::Renderer`:
//   exa::ExaBricks / ScalarField / TriangleMesh / Config  (exa/ExaBricks.h, ScalarField.h,
//       TriangleMesh.h, Config.h) — same static load functions, same file formats
//   exa::Renderer   (exa/OptixRenderer.h:32-97) — same public method set and the public
//       members the viewer touches (frameState, fbSize, worldSpaceBounds, scalarFields)
// The math types stand in for the un-vendored owl::common ones (vec3f, box3f,
// interval, affine3f with xfmPoint/xfmVector/rcp).
#pragma once

#include "../../include/exa_hip.h"

#include <cmath>
#include <limits>
#include <memory>
#include <string>
#include <vector>

namespace exa {

struct vec2i { int x = 0, y = 0; vec2i() = default; vec2i(int a, int b) : x(a), y(b) {} explicit vec2i(int a) : x(a), y(a) {} };
struct vec3i { int x = 0, y = 0, z = 0; vec3i() = default; vec3i(int a, int b, int c) : x(a), y(b), z(c) {} };
struct vec3f {
  float x = 0, y = 0, z = 0;
  vec3f() = default;
  explicit vec3f(float s) : x(s), y(s), z(s) {}
  vec3f(float a, float b, float c) : x(a), y(b), z(c) {}
  explicit vec3f(const vec3i &v) : x(float(v.x)), y(float(v.y)), z(float(v.z)) {}
  float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline vec3f operator+(vec3f a, vec3f b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline vec3f operator-(vec3f a, vec3f b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline vec3f operator*(vec3f a, vec3f b) { return { a.x * b.x, a.y * b.y, a.z * b.z }; }
inline vec3f operator*(float s, vec3f a) { return { s * a.x, s * a.y, s * a.z }; }
inline vec3f operator*(vec3f a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline vec3f operator/(vec3f a, float s) { return { a.x / s, a.y / s, a.z / s }; }
inline vec3f operator-(vec3f a) { return { -a.x, -a.y, -a.z }; }
inline bool operator==(vec3f a, vec3f b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
inline bool operator!=(vec3f a, vec3f b) { return !(a == b); }
inline float dot(vec3f a, vec3f b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float length(vec3f a) { return std::sqrt(dot(a, a)); }
inline vec3f normalize(vec3f a) { return (1.f / std::sqrt(dot(a, a))) * a; }
inline vec3f cross(vec3f a, vec3f b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }

template <typename T> struct interval {
  T lower = std::numeric_limits<T>::infinity(), upper = -std::numeric_limits<T>::infinity();
  interval() = default;
  interval(T lo, T hi) : lower(lo), upper(hi) {}
  void extend(T v) { lower = std::fmin(lower, v); upper = std::fmax(upper, v); }
};
typedef interval<float> range1f;

struct box3f {
  vec3f lower{ std::numeric_limits<float>::infinity() }, upper{ -std::numeric_limits<float>::infinity() };
  box3f() = default;
  box3f(vec3f lo, vec3f hi) : lower(lo), upper(hi) {}
  void extend(vec3f p)
  {
    lower = { std::fmin(lower.x, p.x), std::fmin(lower.y, p.y), std::fmin(lower.z, p.z) };
    upper = { std::fmax(upper.x, p.x), std::fmax(upper.y, p.y), std::fmax(upper.z, p.z) };
  }
  void extend(const box3f &b) { extend(b.lower); extend(b.upper); }
  vec3f center() const { return 0.5f * (lower + upper); }
  vec3f span() const { return upper - lower; }
};

struct linear3f { vec3f vx{ 1, 0, 0 }, vy{ 0, 1, 0 }, vz{ 0, 0, 1 }; };
struct affine3f {
  linear3f l; vec3f p{ 0, 0, 0 };
  static affine3f translate(vec3f t) { affine3f a; a.p = t; return a; }
  static affine3f scale(vec3f s) { affine3f a; a.l.vx = { s.x, 0, 0 }; a.l.vy = { 0, s.y, 0 }; a.l.vz = { 0, 0, s.z }; return a; }
};
inline vec3f xfmVector(const affine3f &a, vec3f v) { return v.x * a.l.vx + (v.y * a.l.vy + v.z * a.l.vz); }
inline vec3f xfmPoint(const affine3f &a, vec3f v) { return v.x * a.l.vx + (v.y * a.l.vy + (v.z * a.l.vz + a.p)); }
affine3f operator*(const affine3f &a, const affine3f &b);
affine3f rcp(const affine3f &a);

static const int NUM_XF_VALUES = EXA_NUM_XF_VALUES;
static const int MAX_CHANNELS = EXA_MAX_CHANNELS;
static const int MAX_ISO_SURFACES = EXA_MAX_ISO_SURFACES;
static const int MAX_CONTOUR_PLANES = EXA_MAX_CONTOUR_PLANES;

// ---- data model (file formats unchanged) ----
struct ExaBricks {
  typedef std::shared_ptr<ExaBricks> SP;
  // all bricks flat: record i = {size.xyz, lower.xyz, level} (the `.bricks` header order)
  std::vector<int32_t> bricks7;
  std::vector<int32_t> cellIDs;      // concatenated per-brick cell ids
  size_t totalNumCells = 0;
  // the reference's build option ALLOW_EMPTY_CELLS (CMakeLists.txt:70-73), here a property of the loaded bricks: cell id -1
  // = no cell (exa/ExaBricks.cpp:46-49); the renderer then gathers the poison value for it and skips such corners
  bool allowEmptyCells = false;
  size_t numBricks() const { return bricks7.size() / 7; }
  static SP load(const std::string &brickFileName);        // exa/ExaBricks.cpp:21-55
  void save(const std::string &brickFileName) const;       // builder/builder.cpp:895-902
  box3f getBounds() const;                                 // exa/ExaBricks.cpp:57-63
};

struct ScalarField {
  typedef std::shared_ptr<ScalarField> SP;
  static SP load(const std::string &fieldName, const std::string &fileName);              // exa/ScalarField.cpp:22-55
  static SP loadAndComputeMagnitude(const std::string &fieldName, const std::string &fnx,
                                    const std::string &fny, const std::string &fnz);      // :57-98
  static SP createFromExpression(const std::string &fieldName, const std::vector<SP> &fields,
                                 const std::vector<std::string> &tokens);                 // :100-226
  std::string name;
  interval<float> valueRange;
  std::vector<float> value;
};

struct TriangleMesh {
  typedef std::shared_ptr<TriangleMesh> SP;
  std::vector<vec3f> vertex;
  std::vector<vec3i> index;
  static std::vector<SP> load(const std::string &fileName);                               // exa/TriangleMesh.cpp:21-71
};

struct Config {
  typedef std::shared_ptr<Config> SP;
  static SP parseConfigFile(const std::string &fileName);                                 // exa/Config.cpp:57-180
  void finalize();                                                                        // :23-44
  box3f getBounds();                                                                      // :48-55
  std::vector<TriangleMesh::SP> surfaces;
  struct {
    ExaBricks::SP sp;
    box3f remap_from{ vec3f(0.f), vec3f(1.f) };
    box3f remap_to{ vec3f(0.f), vec3f(1.f) };
    affine3f voxelSpaceTransform;
  } bricks;
  std::vector<ScalarField::SP> scalarFields;
};

// programs/FrameState.h:29-71 with the member names the viewer writes to
struct FrameState {
  struct { vec3f pos, dir00, dirDu, dirDv; } camera;
  struct { bool enabled = false; float value = 0.f; int channel = 0; } isoSurface[MAX_ISO_SURFACES];
  struct { bool enabled = false; vec3f normal{ 1.f, 0.f, 0.f }; int channel = 0; float offset = .5f; } contourPlane[MAX_CONTOUR_PLANES];
  struct { box3f coords; bool enabled = false; } clipBox;
  struct { float length = 1e20f; bool enabled = true; } ao;
  float clockScale = 0.f;
  affine3f voxelSpaceTransform;
  int frameID = 0;
  interval<float> xfDomain[MAX_CHANNELS];
  float xfOpacityScale = 1.f;
};

// exa::OptixRenderer's interface (exa/OptixRenderer.h:32-97) over the C ABI
struct Renderer {
  typedef std::shared_ptr<Renderer> SP;
  Renderer(ExaBricks::SP input, std::vector<TriangleMesh::SP> surfaces, std::vector<ScalarField::SP> scalarFields,
           int device = 0);
  // one renderer that drives several GPUs of the node (exa_hip_create_multi): image tiles dealt round-robin, every
  // device stores its tiles straight into the frame on devices[0]
  Renderer(ExaBricks::SP input, std::vector<TriangleMesh::SP> surfaces, std::vector<ScalarField::SP> scalarFields,
           const std::vector<int> &devices);
  ~Renderer();
  Renderer(const Renderer &) = delete;

  void setVoxelSpaceTransform(const affine3f &voxelSpaceTransform);
  void resizeFrameBuffer(void *fbPointer, const vec2i &fbSize);      // fbPointer: host memory, owned by the caller
  void updateIsoValues(const float *isoValues, const int *channels, const int *enabled);
  void updateContourPlanes(const vec3f *normals, const float *offsets, const int *channels, const int *enabled);
  void updateCamera(const vec3f &pos, const vec3f &dir00, const vec3f &dirDu, const vec3f &dirDv);
  void updateXF(int chan, const float *opacities, const std::vector<vec3f> &colorMap, const interval<float> &xfDomain,
                float xfOpacityScale = .1f);
  void updateFrameID(int frameID);
  void updateDt(float dt);
  void setSpaceSkipping(bool enable);
  void setGradientShadingDVR(bool enable);
  void setGradientShadingISO(bool enable);
  void setTracerEnabled(bool enable);
  void resetTracer();
  bool advanceTracer();
  void render();
  // the frame into a DEVICE buffer (memory of the first device), queued on `hipStream` without waiting: the caller
  // orders its copy-out behind it on that stream and may start the next frame into another buffer meanwhile
  void renderAsync(void *deviceColorBuffer, void *hipStream);

  // exa_hip_set_option (include/exa_hip.h): tuning knobs of the module that have no counterpart in the reference's
  // interface (which walk of the region kd-tree the march takes, launch order, AO launch plan ...); none changes a picture
  // beyond the stated float tolerance
  void setOption(const std::string &key, int value);

  ExaHipStats stats() const;
  ExaHipStats renderStats();                 // the same frame through the counting variant of the kernels
  size_t numRegions = 0, numLeafEntries = 0; // what Regions::buildFrom produced (exa/Regions.cpp:308-319 prints them)

  std::vector<ScalarField::SP> scalarFields;
  ExaBricks::SP input;
  box3f voxelSpaceBounds, worldSpaceBounds;
  bool multiFieldDvr = true;
  bool gradientShadingDVR = true, gradientShadingISO = true;
  bool doSpaceSkipping = true;
  FrameState frameState;
  vec2i fbSize;
  struct {                                   // exa/OptixRenderer.h:160-170
    int tracerEnabled = false;
    vec3i tracerChannels{ 0, 1, 2 };
    int numTraces = 1000;
    int numTimesteps = 100;
    float steplen = 1e-6f;
    int timestepHost = 0;
    box3f seedRegion{ vec3f(.3f, .3f, .5f), vec3f(.8f, .8f, .5f) };
  } traces;

private:
  void init(std::vector<TriangleMesh::SP> surfaces, const std::vector<int> &devices);
  void pushState();
  ExaPrep *prep = nullptr;
  ExaHipRenderer *handle = nullptr;
  ExaHipParams params{};
  void *fbPointer = nullptr;
};
typedef Renderer OptixRenderer;   // source compatibility for hosts written against the reference

} // namespace exa
