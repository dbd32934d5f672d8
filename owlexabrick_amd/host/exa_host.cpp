// exa_host.cpp — loaders for the reference's file formats and the exa::Renderer facade.
#include "exa_host.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <random>
#include <stack>
#include <stdexcept>

namespace exa {

// ------------------------------------------------------------------ math
affine3f operator*(const affine3f &a, const affine3f &b)
{
  affine3f r;
  r.l.vx = xfmVector(a, b.l.vx);
  r.l.vy = xfmVector(a, b.l.vy);
  r.l.vz = xfmVector(a, b.l.vz);
  r.p = xfmPoint(a, b.p);
  return r;
}

affine3f rcp(const affine3f &a)
{
  // inverse of the linear part by adjugate / determinant, then p' = -L^-1 p
  const vec3f c0 = cross(a.l.vy, a.l.vz), c1 = cross(a.l.vz, a.l.vx), c2 = cross(a.l.vx, a.l.vy);
  const float det = dot(a.l.vx, c0);
  affine3f r;
  r.l.vx = vec3f(c0.x, c1.x, c2.x) / det;
  r.l.vy = vec3f(c0.y, c1.y, c2.y) / det;
  r.l.vz = vec3f(c0.z, c1.z, c2.z) / det;
  r.p = -xfmVector(r, a.p);
  return r;
}

// ------------------------------------------------------------------ ExaBricks
// `.bricks`: per brick int32 size[3], lower[3], level, then size.x*size.y*size.z int32 cell ids,
// x fastest, no header (builder/builder.cpp:895-902, exa/ExaBricks.cpp:27-38)
ExaBricks::SP ExaBricks::load(const std::string &fileName)
{
  std::ifstream in(fileName, std::ios::binary);
  if (!in.good()) throw std::runtime_error("could not open " + fileName);
  auto exa = std::make_shared<ExaBricks>();
  for (;;) {
    int32_t rec[7];
    in.read(reinterpret_cast<char *>(rec), sizeof(rec));
    if (!in.good()) break;
    const size_t n = size_t(rec[0]) * size_t(rec[1]) * size_t(rec[2]);
    if (rec[0] <= 0 || rec[1] <= 0 || rec[2] <= 0 || rec[6] < 0 || rec[6] > 30)
      throw std::runtime_error("malformed brick record in " + fileName);
    exa->bricks7.insert(exa->bricks7.end(), rec, rec + 7);
    const size_t at = exa->cellIDs.size();
    exa->cellIDs.resize(at + n);
    in.read(reinterpret_cast<char *>(exa->cellIDs.data() + at), std::streamsize(n * sizeof(int32_t)));
    if (size_t(in.gcount()) != n * sizeof(int32_t)) throw std::runtime_error("truncated brick file " + fileName);
    exa->totalNumCells += n;
  }
  return exa;
}

void ExaBricks::save(const std::string &fileName) const
{
  std::ofstream out(fileName, std::ios::binary);
  if (!out.good()) throw std::runtime_error("could not open " + fileName);
  size_t at = 0;
  for (size_t b = 0; b < numBricks(); b++) {
    const int32_t *r = &bricks7[7 * b];
    const size_t n = size_t(r[0]) * size_t(r[1]) * size_t(r[2]);
    out.write(reinterpret_cast<const char *>(r), 7 * sizeof(int32_t));
    out.write(reinterpret_cast<const char *>(cellIDs.data() + at), std::streamsize(n * sizeof(int32_t)));
    at += n;
  }
}

box3f ExaBricks::getBounds() const
{
  box3f b;
  for (size_t i = 0; i < numBricks(); i++) {
    const int32_t *r = &bricks7[7 * i];
    const vec3i lo(r[3], r[4], r[5]);
    b.extend(vec3f(lo));
    b.extend(vec3f(vec3i(lo.x + r[0] * (1 << r[6]), lo.y + r[1] * (1 << r[6]), lo.z + r[2] * (1 << r[6]))));
  }
  return b;
}

// ------------------------------------------------------------------ ScalarField
ScalarField::SP ScalarField::load(const std::string &fieldName, const std::string &fileName)
{
  std::ifstream f(fileName, std::ios::binary);
  if (!f.good()) throw std::runtime_error("error in opening scalar file " + fileName);
  f.seekg(0, f.end);
  // The reference sizes the vector by the file's BYTE count (exa/ScalarField.cpp:27-34): four times
  // too many elements, the tail stays 0.0f and is folded into valueRange.  Kept: valueRange is the
  // viewer's default transfer-function domain, and cell-id bounds checks use this size.
  const size_t numElements = size_t(f.tellg());
  f.seekg(0, f.beg);
  auto sf = std::make_shared<ScalarField>();
  sf->name = fieldName;
  sf->value.resize(numElements);
  f.read(reinterpret_cast<char *>(sf->value.data()), std::streamsize(numElements));
  for (float v : sf->value) sf->valueRange.extend(v);
  return sf;
}

ScalarField::SP ScalarField::loadAndComputeMagnitude(const std::string &fieldName, const std::string &fnx,
                                                     const std::string &fny, const std::string &fnz)
{
  auto x = load(fieldName + ".x", fnx), y = load(fieldName + ".y", fny), z = load(fieldName + ".z", fnz);
  if (x->value.size() != y->value.size() || x->value.size() != z->value.size())
    throw std::runtime_error("vector field components differ in size");
  auto sf = std::make_shared<ScalarField>();
  sf->name = fieldName;
  sf->value.resize(x->value.size());
  for (size_t i = 0; i < sf->value.size(); i++) {
    sf->value[i] = length(vec3f(x->value[i], y->value[i], z->value[i]));
    sf->valueRange.extend(sf->value[i]);
  }
  return sf;
}

ScalarField::SP ScalarField::createFromExpression(const std::string &fieldName, const std::vector<SP> &fields,
                                                  const std::vector<std::string> &tokens)
{
  // postfix expression over the already-loaded fields: %N placeholders, select, binary and unary
  // operators, float constants (exa/ScalarField.cpp:100-226).  Tokens are compiled once.
  enum Op { PUSH_FIELD, PUSH_CONST, SELECT, ADD, SUB, MUL, DIV, POW, EQ, NE, LT, GT, LE, GE, LOG, ABS, SQRT };
  struct Ins { Op op; unsigned field; float c; };
  std::vector<Ins> prog;
  if (fields.empty()) throw std::runtime_error("expression field needs at least one loaded field");
  for (std::string t : tokens) {
    auto strip = [&](char c) {
      while (!t.empty() && t.front() == c) t.erase(t.begin());
      while (!t.empty() && t.back() == c) t.pop_back();
    };
    strip('"'); strip(' ');
    if (t.empty()) throw std::runtime_error("empty token in expression '" + fieldName + "'");
    static const struct { const char *s; Op op; } table[] = {
      { "select", SELECT }, { "+", ADD }, { "-", SUB }, { "*", MUL }, { "/", DIV }, { "**", POW }, { "==", EQ },
      { "!=", NE }, { "<", LT }, { ">", GT }, { "<=", LE }, { ">=", GE }, { "log", LOG }, { "abs", ABS }, { "sqrt", SQRT } };
    bool done = false;
    if (t[0] == '%') {
      const unsigned f = (unsigned)std::stoi(t.substr(t.find_first_not_of('%')));
      if (f >= fields.size()) throw std::runtime_error("invalid placeholder token: " + t);
      prog.push_back({ PUSH_FIELD, f, 0.f });
      done = true;
    }
    for (auto &e : table) if (!done && t == e.s) { prog.push_back({ e.op, 0, 0.f }); done = true; }
    if (!done) prog.push_back({ PUSH_CONST, 0, (float)std::stod(t) });
  }
  auto sf = std::make_shared<ScalarField>();
  sf->name = fieldName;
  sf->value.resize(fields[0]->value.size());
  std::vector<float> st;
  for (size_t i = 0; i < sf->value.size(); i++) {
    st.clear();
    for (const Ins &in : prog) {
      auto pop = [&]() { if (st.empty()) throw std::runtime_error("invalid expression"); float v = st.back(); st.pop_back(); return v; };
      switch (in.op) {
        case PUSH_FIELD: st.push_back(fields[in.field]->value[i]); break;
        case PUSH_CONST: st.push_back(in.c); break;
        case SELECT: { float b = pop(), a = pop(); int m = (int)pop(); st.push_back(m ? a : b); break; }
        case LOG: st.push_back(std::log(pop())); break;
        case ABS: st.push_back(std::fabs(pop())); break;
        case SQRT: st.push_back(std::sqrt(pop())); break;
        default: {
          const float b = pop(), a = pop();
          switch (in.op) {
            case ADD: st.push_back(a + b); break; case SUB: st.push_back(a - b); break;
            case MUL: st.push_back(a * b); break; case DIV: st.push_back(a / b); break;
            case POW: st.push_back(std::pow(a, b)); break;
            case EQ: st.push_back(a == b); break; case NE: st.push_back(a != b); break;
            case LT: st.push_back(a < b); break;  case GT: st.push_back(a > b); break;
            case LE: st.push_back(a <= b); break; default: st.push_back(a >= b); break;
          }
        }
      }
    }
    if (st.size() != 1) throw std::runtime_error("invalid expression");
    sf->value[i] = st.back();
    sf->valueRange.extend(sf->value[i]);
  }
  return sf;
}

// ------------------------------------------------------------------ TriangleMesh
std::vector<TriangleMesh::SP> TriangleMesh::load(const std::string &fileName)
{
  std::ifstream in(fileName, std::ios::binary);
  if (!in.good()) throw std::runtime_error("cannot open file");
  std::vector<SP> result;
  for (;;) {
    int32_t nv = 0, nt = 0;
    in.read(reinterpret_cast<char *>(&nv), sizeof(nv));
    if (!in.good() || nv < 0) break;
    auto m = std::make_shared<TriangleMesh>();
    m->vertex.resize(nv);
    in.read(reinterpret_cast<char *>(m->vertex.data()), std::streamsize(nv * sizeof(vec3f)));
    in.read(reinterpret_cast<char *>(&nt), sizeof(nt));
    if (!in.good() || nt < 0) throw std::runtime_error("broken triangle model");
    m->index.resize(nt);
    in.read(reinterpret_cast<char *>(m->index.data()), std::streamsize(nt * sizeof(vec3i)));
    for (const vec3i &t : m->index)
      if (t.x < 0 || t.y < 0 || t.z < 0 || t.x >= nv || t.y >= nv || t.z >= nv) throw std::runtime_error("broken triangle model");
    result.push_back(m);
  }
  return result;
}

// ------------------------------------------------------------------ Config
void Config::finalize()
{
  const affine3f voxelCS = affine3f::translate(bricks.remap_from.lower) * affine3f::scale(bricks.remap_from.span());
  const affine3f worldCS = affine3f::translate(bricks.remap_to.lower) * affine3f::scale(bricks.remap_to.span());
  bricks.voxelSpaceTransform = voxelCS * rcp(worldCS);
}

box3f Config::getBounds()
{
  box3f b = bricks.sp->getBounds();
  const affine3f inv = rcp(bricks.voxelSpaceTransform);
  return box3f(xfmPoint(inv, b.lower), xfmPoint(inv, b.upper));
}

Config::SP Config::parseConfigFile(const std::string &fileName)
{
  std::ifstream file(fileName);
  if (!file.good()) throw std::runtime_error("error in opening config file '" + fileName + "'");
  // whitespace-separated tokens; '#' starts a comment that runs to the end of the line (:72-78)
  std::vector<std::string> tok;
  std::string line;
  while (std::getline(file, line)) {
    size_t i = 0;
    while (i < line.size()) {
      while (i < line.size() && std::strchr(" \t\n\r", line[i])) i++;
      if (i >= line.size() || line[i] == '#') break;
      size_t j = i;
      while (j < line.size() && !std::strchr(" \t\n\r", line[j])) j++;
      tok.push_back(line.substr(i, j - i));
      i = j;
    }
  }
  auto config = std::make_shared<Config>();
  const size_t slash = fileName.rfind('/');
  const std::string base = slash == std::string::npos ? std::string("./") : fileName.substr(0, slash) + "/";
  size_t it = 0;
  auto need = [&](size_t n) {
    if (it + n >= tok.size()) throw std::runtime_error("error in parsing config file: missing arguments after '" + tok[it] + "'");
  };
  auto f = [&](size_t k) { return std::stof(tok[it + k]); };
  while (it < tok.size()) {
    const std::string &t = tok[it];
    if (t == "remap_from" || t == "remap_to") {
      need(6);
      box3f &b = t == "remap_from" ? config->bricks.remap_from : config->bricks.remap_to;
      b.lower = vec3f(f(1), f(2), f(3));
      b.upper = vec3f(f(4), f(5), f(6));
      it += 7;
    } else if (t == "scalar") {
      need(2);
      const std::string name = tok[it + 1];
      if (tok[it + 2] == "expr") {
        it += 3;
        std::vector<std::string> expr;
        for (;;) {
          if (it >= tok.size()) throw std::runtime_error("unterminated expression for scalar '" + name + "'");
          expr.push_back(tok[it]);
          const bool last = tok[it].back() == '"';
          it++;
          if (last) break;
        }
        config->scalarFields.push_back(ScalarField::createFromExpression(name, config->scalarFields, expr));
      } else {
        config->scalarFields.push_back(ScalarField::load(name, base + tok[it + 2]));
        it += 3;
      }
    } else if (t == "vector") {
      need(4);
      config->scalarFields.push_back(ScalarField::loadAndComputeMagnitude(tok[it + 1], base + tok[it + 2], base + tok[it + 3], base + tok[it + 4]));
      it += 5;
    } else if (t == "value_range") {
      need(2);
      if (config->scalarFields.empty()) throw std::runtime_error("value_range before any scalar field");
      config->scalarFields.back()->valueRange = interval<float>(f(1), f(2));
      it += 3;
    } else if (t == "bricks") {
      need(1);
      config->bricks.sp = ExaBricks::load(base + tok[it + 1]);
      it += 2;
    } else if (t == "triangles") {
      need(1);
      config->surfaces = TriangleMesh::load(base + tok[it + 1]);
      it += 2;
    } else {
      throw std::runtime_error("error in parsing config file: unknown token '" + t + "'");
    }
  }
  config->finalize();
  return config;
}

// ------------------------------------------------------------------ Renderer
static void check(int rc, ExaHipRenderer *h)
{
  if (rc) throw std::runtime_error(exa_hip_last_error(h));
}

Renderer::Renderer(ExaBricks::SP in, std::vector<TriangleMesh::SP> surfaces, std::vector<ScalarField::SP> fields, int device)
  : scalarFields(fields), input(in)
{ init(surfaces, std::vector<int>(1, device)); }

Renderer::Renderer(ExaBricks::SP in, std::vector<TriangleMesh::SP> surfaces, std::vector<ScalarField::SP> fields,
                   const std::vector<int> &devices)
  : scalarFields(fields), input(in)
{ init(surfaces, devices); }

void Renderer::init(std::vector<TriangleMesh::SP> surfaces, const std::vector<int> &devices)
{
  ExaBricks::SP in = input;
  std::vector<ScalarField::SP> &fields = scalarFields;
  if (devices.empty()) throw std::runtime_error("no device");
  if (!in || in->numBricks() == 0) throw std::runtime_error("no bricks");
  if (fields.empty() || (int)fields.size() > MAX_CHANNELS) throw std::runtime_error("1..10 scalar fields required");
  voxelSpaceBounds = in->getBounds();
  std::vector<const float *> ptr;
  std::vector<uint64_t> len;
  for (auto &f : fields) { ptr.push_back(f->value.data()); len.push_back(f->value.size()); }
  const int nRegionFields = multiFieldDvr ? (int)fields.size() : 1;                  // OptixRenderer.cpp:154
  if (exa_prep_create_ex(in->bricks7.data(), in->numBricks(), in->cellIDs.data(), in->cellIDs.size(),
                         ptr.data(), len.data(), (int)fields.size(), nRegionFields, 0,
                         in->allowEmptyCells ? EXA_PREP_ALLOW_EMPTY_CELLS : 0, &prep))
    throw std::runtime_error(exa_prep_last_error());
  ExaHipScene scene;
  exa_prep_scene(prep, &scene);
  numRegions = scene.numRegions; numLeafEntries = scene.leafListSize;
  std::vector<int32_t> devs(devices.begin(), devices.end());
  if (devs.size() == 1 ? exa_hip_create(&scene, devs[0], &handle)
                       : exa_hip_create_multi(&scene, devs.data(), (int32_t)devs.size(), &handle)) {
    const std::string msg = exa_hip_last_error(nullptr);
    exa_prep_destroy(prep);
    prep = nullptr;
    throw std::runtime_error(msg);
  }
  // createSurfaces (OptixRenderer.cpp:554-612): all meshes go into one triangle set
  std::vector<float> verts;
  std::vector<int32_t> tris;
  for (auto &m : surfaces) {
    const int32_t base = (int32_t)(verts.size() / 3);
    for (const vec3f &v : m->vertex) { verts.push_back(v.x); verts.push_back(v.y); verts.push_back(v.z); }
    for (const vec3i &t : m->index) { tris.push_back(base + t.x); tris.push_back(base + t.y); tris.push_back(base + t.z); }
  }
  if (!tris.empty() && exa_hip_set_triangles(handle, verts.data(), verts.size() / 3, tris.data(), tris.size() / 3)) {
    const std::string msg = exa_hip_last_error(handle);
    exa_hip_destroy(handle); exa_prep_destroy(prep);
    handle = nullptr; prep = nullptr;
    throw std::runtime_error(msg);
  }
  if (fields.size() >= 3) resetTracer();          // the constructor's resetTracer() (:288); needs three velocity fields
  params.dt = 0.5f;
  params.numPrimaryChannels = multiFieldDvr ? (int)fields.size() : 1;                 // :284
  params.colormapChannel = (!multiFieldDvr && fields.size() > 1) ? 1 : 0;             // :278-283
  params.numChannels = params.numPrimaryChannels;                                     // :650
  worldSpaceBounds = voxelSpaceBounds;
}

Renderer::~Renderer()
{
  if (handle) exa_hip_destroy(handle);
  if (prep) exa_prep_destroy(prep);
}

void Renderer::setVoxelSpaceTransform(const affine3f &x)
{
  frameState.voxelSpaceTransform = x;
  const affine3f inv = rcp(x);                                                        // OptixRenderer.cpp:330-332
  worldSpaceBounds = box3f(xfmPoint(inv, voxelSpaceBounds.lower), xfmPoint(inv, voxelSpaceBounds.upper));
}

void Renderer::resizeFrameBuffer(void *fb, const vec2i &size)
{
  fbSize = size;
  fbPointer = fb;
  check(exa_hip_resize(handle, size.x, size.y), handle);
}

void Renderer::updateIsoValues(const float *v, const int *ch, const int *en)
{
  for (int i = 0; i < MAX_ISO_SURFACES; i++) {
    frameState.isoSurface[i].value = v[i];
    frameState.isoSurface[i].channel = ch[i];
    frameState.isoSurface[i].enabled = en[i] != 0;
  }
}

void Renderer::updateContourPlanes(const vec3f *n, const float *off, const int *ch, const int *en)
{
  for (int i = 0; i < MAX_CONTOUR_PLANES; i++) {
    frameState.contourPlane[i].normal = normalize(n[i]);
    frameState.contourPlane[i].offset = off[i];
    frameState.contourPlane[i].channel = ch[i];
    frameState.contourPlane[i].enabled = en[i] != 0;
  }
}

void Renderer::updateCamera(const vec3f &pos, const vec3f &dir00, const vec3f &dirDu, const vec3f &dirDv)
{
  frameState.camera.pos = pos; frameState.camera.dir00 = dir00;
  frameState.camera.dirDu = dirDu; frameState.camera.dirDv = dirDv;
}

void Renderer::updateXF(int chan, const float *opacities, const std::vector<vec3f> &colorMap,
                        const interval<float> &xfDomain, float xfOpacityScale)
{
  if (colorMap.size() != (size_t)NUM_XF_VALUES) throw std::runtime_error("mismatching xf size!?");   // :382-383
  frameState.xfDomain[chan] = xfDomain;
  frameState.xfOpacityScale = xfOpacityScale;
  float lut[NUM_XF_VALUES][4];
  for (int i = 0; i < NUM_XF_VALUES; i++) {
    lut[i][0] = colorMap[i].x; lut[i][1] = colorMap[i].y; lut[i][2] = colorMap[i].z; lut[i][3] = opacities[i];
  }
  check(exa_hip_set_xf(handle, chan, &lut[0][0]), handle);
}

void Renderer::updateFrameID(int id) { frameState.frameID = id; }
void Renderer::updateDt(float dt) { params.dt = dt; }
void Renderer::setSpaceSkipping(bool e) { doSpaceSkipping = e; }
void Renderer::setGradientShadingDVR(bool e) { gradientShadingDVR = e; }
void Renderer::setGradientShadingISO(bool e) { gradientShadingISO = e; }
void Renderer::setTracerEnabled(bool e)
{
  traces.tracerEnabled = e;
  check(exa_hip_set_tracer_enabled(handle, e), handle);
}

// OptixRenderer::resetTracer (exa/OptixRenderer.cpp:450-472): seeds drawn inside seedRegion * size
void Renderer::resetTracer()
{
  const vec3f size = voxelSpaceBounds.upper - voxelSpaceBounds.lower;
  std::default_random_engine engine(0);
  std::uniform_real_distribution<float> x(traces.seedRegion.lower.x * size.x, traces.seedRegion.upper.x * size.x);
  std::uniform_real_distribution<float> y(traces.seedRegion.lower.y * size.y, traces.seedRegion.upper.y * size.y);
  std::uniform_real_distribution<float> z(traces.seedRegion.lower.z * size.z, traces.seedRegion.upper.z * size.z);
  std::vector<float> seeds(size_t(traces.numTraces) * 3);
  for (int i = 0; i < traces.numTraces; ++i) { seeds[3 * i] = x(engine); seeds[3 * i + 1] = y(engine); seeds[3 * i + 2] = z(engine); }
  ExaHipTracer t;
  t.enabled = traces.tracerEnabled;
  t.channels[0] = traces.tracerChannels.x; t.channels[1] = traces.tracerChannels.y; t.channels[2] = traces.tracerChannels.z;
  t.numTraces = traces.numTraces; t.numTimesteps = traces.numTimesteps; t.steplen = traces.steplen;
  check(exa_hip_reset_tracer(handle, &t, seeds.data()), handle);
  traces.timestepHost = 0;
}

bool Renderer::advanceTracer()
{
  if (!traces.tracerEnabled) return false;
  int32_t rebuild = 0;
  check(exa_hip_advance_tracer(handle, &rebuild), handle);
  traces.timestepHost++;
  return rebuild != 0;
}

void Renderer::pushState()
{
  ExaHipFrameState fs;
  std::memset(&fs, 0, sizeof(fs));
  auto put = [](float *d, const vec3f &v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; };
  put(fs.cam_pos, frameState.camera.pos); put(fs.cam_dir00, frameState.camera.dir00);
  put(fs.cam_dirDu, frameState.camera.dirDu); put(fs.cam_dirDv, frameState.camera.dirDv);
  bool contour = false;
  for (int i = 0; i < MAX_ISO_SURFACES; i++) {
    fs.iso[i].enabled = frameState.isoSurface[i].enabled;
    fs.iso[i].value = frameState.isoSurface[i].value;
    fs.iso[i].channel = frameState.isoSurface[i].channel;
  }
  for (int i = 0; i < MAX_CONTOUR_PLANES; i++) {
    fs.contour[i].enabled = frameState.contourPlane[i].enabled;
    put(fs.contour[i].normal, frameState.contourPlane[i].normal);
    fs.contour[i].channel = frameState.contourPlane[i].channel;
    fs.contour[i].offset = frameState.contourPlane[i].offset;
    contour |= frameState.contourPlane[i].enabled;
  }
  put(fs.clipBox.lo, frameState.clipBox.coords.lower); put(fs.clipBox.hi, frameState.clipBox.coords.upper);
  fs.clipBox.enabled = frameState.clipBox.enabled;
  fs.ao.length = frameState.ao.length; fs.ao.enabled = frameState.ao.enabled;
  fs.clockScale = frameState.clockScale;
  put(fs.xfm_vx, frameState.voxelSpaceTransform.l.vx); put(fs.xfm_vy, frameState.voxelSpaceTransform.l.vy);
  put(fs.xfm_vz, frameState.voxelSpaceTransform.l.vz); put(fs.xfm_p, frameState.voxelSpaceTransform.p);
  fs.frameID = frameState.frameID;
  for (int c = 0; c < MAX_CHANNELS; c++) { fs.xfDomain[c][0] = frameState.xfDomain[c].lower; fs.xfDomain[c][1] = frameState.xfDomain[c].upper; }
  fs.xfOpacityScale = frameState.xfOpacityScale;
  params.gradientShadingDVR = gradientShadingDVR;
  params.gradientShadingISO = gradientShadingISO;
  params.spaceSkippingEnabled = !contour && doSpaceSkipping;                          // OptixRenderer.cpp:418-432
  check(exa_hip_set_frame_state(handle, &fs), handle);
  check(exa_hip_set_params(handle, &params), handle);
}

void Renderer::render()
{
  if (!fbPointer) throw std::runtime_error("resizeFrameBuffer() has not been called");
  pushState();
  check(exa_hip_render(handle, static_cast<uint32_t *>(fbPointer), 0, nullptr, 0), handle);
}

void Renderer::renderAsync(void *deviceColorBuffer, void *hipStream)
{
  pushState();
  check(exa_hip_render(handle, static_cast<uint32_t *>(deviceColorBuffer), 1, hipStream, 1), handle);
}

ExaHipStats Renderer::renderStats()
{
  if (!fbPointer) throw std::runtime_error("resizeFrameBuffer() has not been called");
  pushState();
  ExaHipStats s;
  check(exa_hip_render_stats(handle, static_cast<uint32_t *>(fbPointer), 0, &s), handle);
  return s;
}

void Renderer::setOption(const std::string &key, int value) { check(exa_hip_set_option(handle, key.c_str(), value), handle); }

ExaHipStats Renderer::stats() const
{
  ExaHipStats s;
  exa_hip_get_stats(handle, &s);
  return s;
}

} // namespace exa
