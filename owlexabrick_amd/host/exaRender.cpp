// exaRender — headless stand-in for the reference's exaViewer (exa/viewer.cpp): same config
// file, same camera / transfer-function / iso flags, same per-frame call sequence into the
// renderer, N frames instead of a GLUT loop, result written as binary PPM.
//   exaRender cfg.exa [--size W H] [--camera px py pz  ix iy iz  ux uy uz] [--fov deg]
//             [--dt f] [--xf file.xf] [--xf-scale s] [--range lo hi] [--isovals a b] [--isochans a b]
//             [--contourplane nx ny nz offset]... [--contourchan c]...   (exa/viewer.cpp:1199-1207, at most 3 planes)
//             [--clip-box lx ly lz ux uy uz] [--ao] [--ao-length l] [--no-pg] [--no-space-skipping]
//             [--gradientShadingDVR 0|1] [--gradientShadingISO 0|1] [--frames N] [-o out.ppm] [--info] [--stats]
//             [--gpus N | --devices 0,1,..]  one renderer over several GPUs (tiles dealt round-robin, stored straight into
//                                            the first device's frame); a device may be listed more than once
//             [--allow-empty-cells]          cell id -1 in the .bricks file = no cell (the reference's ALLOW_EMPTY_CELLS build)
//             [--option key=int]...          exa_hip_set_option, e.g. walk=1|2 (stack / rope walk of the region kd-tree), ao_overlap=0
//             [--pipeline]                   frames go to two device buffers in turn, frame k's copy to the host overlaps
//                                            frame k+1's march
#include "exa_host.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

using namespace exa;

namespace {
// glutViewer/Camera.cpp:94-120 + glutViewer/OWLViewer.cpp:69-109 + exa/viewer.cpp:226-238
void setupCamera(Renderer &r, vec3f origin, vec3f interest, vec3f up, float fovy, vec2i size)
{
  vec3f vz = (interest == origin) ? vec3f(0, 0, 1) : -normalize(interest - origin);
  vec3f vx = cross(up, vz);
  vx = dot(vx, vx) < 1e-8f ? vec3f(0, 1, 0) : normalize(vx);
  vec3f vy = normalize(cross(vz, vx));
  const float focal = length(interest - origin);
  if (std::fabs(dot(vz, up)) >= 1e-6f) { vx = normalize(cross(up, vz)); vy = normalize(cross(vz, vx)); }
  auto eps = [](vec3f v) { return std::fmax(std::fmax(std::fabs(v.x), std::fabs(v.y)), std::fabs(v.z)) * float(1. / (1 << 21)); };
  const float fd = std::fmax(std::fmax(eps(origin), eps(vx)), focal);
  const float screen_height = 2.f * tanf(fovy / 2.f * (float)M_PI / 180.f) * fd;
  const vec3f vertical = screen_height * vy;
  const vec3f horizontal = (screen_height * (size.x / float(size.y))) * vx;
  const vec3f lower_left = (-fd) * vz - 0.5f * vertical - 0.5f * horizontal;
  const vec3f du = horizontal / float(size.x), dv = vertical / float(size.y);
  std::printf("camera %.9g %.9g %.9g  %.9g %.9g %.9g  %.9g %.9g %.9g  %.9g %.9g %.9g\n", origin.x, origin.y, origin.z,
              lower_left.x, lower_left.y, lower_left.z, du.x, du.y, du.z, dv.x, dv.y, dv.z);
  r.updateCamera(origin, lower_left, du, dv);
}
} // namespace

int main(int argc, char **argv)
{
  try {
    std::string cfgName, outName;
    vec2i size(600, 400);                                         // viewer default window
    vec3f vp, vi, vu;                                             // --camera
    float fov = 70.f, dt = 0.5f, xfScale = 1.f, aoLength = 1e20f;
    float isoVals[2] = { 0, 0 }; int isoChans[2] = { 0, 0 }, isoOn[2] = { 0, 0 };
    bool haveRange = false, ao = false, pg = true, skipping = true, gradDVR = true, gradISO = true, info = false, clip = false, stats = false;
    float range[2] = { 0, 1 }; box3f clipBox;
    std::string xfFile;
    int frames = 1;
    std::vector<int> devices;
    bool pipeline = false;
    bool allowEmptyCells = false;
    std::vector<float> contourPlanes;                             // 4 floats per --contourplane (normal, offset)
    std::vector<std::pair<std::string, int>> options;             // --option key=value
    std::vector<int> contourChans;
    for (int i = 1; i < argc; i++) {
      const std::string a = argv[i];
      auto f = [&]() { if (i + 1 >= argc) throw std::runtime_error("missing value after " + a); return (float)atof(argv[++i]); };
      if (a == "--size" || a == "-win") { size.x = (int)f(); size.y = (int)f(); }
      else if (a == "--camera") { vp = { f(), f(), f() }; vi = { f(), f(), f() }; vu = { f(), f(), f() }; }
      else if (a == "--fov") fov = f();
      else if (a == "--dt") dt = f();
      else if (a == "--xf") { if (i + 1 >= argc) throw std::runtime_error("missing file after --xf"); xfFile = argv[++i]; }
      else if (a == "--xf-scale") xfScale = f();
      else if (a == "--range") { range[0] = f(); range[1] = f(); haveRange = true; }
      else if (a == "--isovals") { isoVals[0] = f(); isoVals[1] = f(); isoOn[0] = isoOn[1] = 1; }
      else if (a == "--isochans") { isoChans[0] = (int)f(); isoChans[1] = (int)f(); }
      else if (a == "--contourplane") { for (int k = 0; k < 4; k++) contourPlanes.push_back(f()); }   // viewer.cpp:1199-1205
      else if (a == "--contourchan") contourChans.push_back((int)f());                                // viewer.cpp:1206-1207
      else if (a == "--clip-box") { clipBox.lower = { f(), f(), f() }; clipBox.upper = { f(), f(), f() }; clip = true; }
      else if (a == "--ao") ao = true;
      else if (a == "--ao-length") aoLength = f();
      else if (a == "--no-pg") pg = false;
      else if (a == "--no-space-skipping") skipping = false;
      else if (a == "--gradientShadingDVR") gradDVR = f() != 0.f;
      else if (a == "--gradientShadingISO") gradISO = f() != 0.f;
      else if (a == "--frames") frames = (int)f();
      else if (a == "-o") { if (i + 1 >= argc) throw std::runtime_error("missing file after -o"); outName = argv[++i]; }
      else if (a == "--info") info = true;
      else if (a == "--stats") stats = true;
      else if (a == "--gpus") { const int n = (int)f(); devices.clear(); for (int d = 0; d < n; d++) devices.push_back(d); }
      else if (a == "--devices") {
        if (i + 1 >= argc) throw std::runtime_error("missing list after --devices");
        devices.clear();
        for (const char *p = argv[++i]; *p;) {
          char *end = nullptr;
          const long d = strtol(p, &end, 10);
          if (end == p || d < 0 || d > 1023 || (*end != ',' && *end != 0)) throw std::runtime_error("--devices wants a comma-separated list of device indices");
          devices.push_back((int)d);
          p = *end == ',' ? end + 1 : end;
          if (*end == ',' && !*p) throw std::runtime_error("--devices wants a comma-separated list of device indices");
        }
        if (devices.empty()) throw std::runtime_error("--devices wants a comma-separated list of device indices");
      }
      else if (a == "--pipeline") pipeline = true;
      else if (a == "--allow-empty-cells") allowEmptyCells = true;    // the reference built with -DALLOW_EMPTY_CELLS=1
      else if (a == "--option") {                                     // exa_hip_set_option: --option walk=2, --option ao_overlap=0 ...
        if (i + 1 >= argc) throw std::runtime_error("missing key=value after --option");
        const std::string kv = argv[++i];
        const size_t eq = kv.find('=');
        char *end = nullptr;
        const long v = eq == std::string::npos ? 0 : std::strtol(kv.c_str() + eq + 1, &end, 10);
        if (eq == std::string::npos || eq == 0 || end == kv.c_str() + eq + 1 || *end) throw std::runtime_error("--option wants key=integer");
        options.emplace_back(kv.substr(0, eq), (int)v);
      }
      else if (a[0] != '-') cfgName = a;
      else throw std::runtime_error("unknown flag " + a);
    }
    if (cfgName.empty()) throw std::runtime_error("usage: exaRender cfg.exa [flags]");
    Config::SP config = Config::parseConfigFile(cfgName);
    if (!config->bricks.sp) throw std::runtime_error("no bricks file specified");
    config->bricks.sp->allowEmptyCells = allowEmptyCells;
    const box3f bounds = config->getBounds();
    std::printf("bricks %zu cells %zu fields %zu\n", config->bricks.sp->numBricks(), config->bricks.sp->totalNumCells,
                config->scalarFields.size());
    for (auto &sf : config->scalarFields)
      std::printf("field %s range %.9g %.9g elements %zu\n", sf->name.c_str(), sf->valueRange.lower, sf->valueRange.upper, sf->value.size());
    std::printf("bounds %.9g %.9g %.9g  %.9g %.9g %.9g\n", bounds.lower.x, bounds.lower.y, bounds.lower.z,
                bounds.upper.x, bounds.upper.y, bounds.upper.z);
    if (info) return 0;

    std::vector<uint32_t> fb(size_t(size.x) * size.y);
    if (devices.empty()) devices.push_back(0);
    Renderer renderer(config->bricks.sp, config->surfaces, config->scalarFields, devices);  // viewer.cpp:1256-1260
    for (const auto &kv : options) renderer.setOption(kv.first, kv.second);
    if (devices.size() > 1) std::printf("devices %zu\n", devices.size());
    renderer.setVoxelSpaceTransform(config->bricks.voxelSpaceTransform);
    renderer.resizeFrameBuffer(fb.data(), size);
    if (vu != vec3f(0.f)) setupCamera(renderer, vp, vi, vu, fov, size);
    else setupCamera(renderer, bounds.center() + vec3f(-.3f, .7f, 1.f) * bounds.span(), bounds.center(), vec3f(0, 1, 0), 70.f, size);

    // default transfer function: alpha ramp, grey ramp colour (the PNG colormaps are UI assets)
    std::vector<float> alpha(NUM_XF_VALUES);
    std::vector<vec3f> color(NUM_XF_VALUES);
    for (int i = 0; i < NUM_XF_VALUES; i++) { alpha[i] = i / float(NUM_XF_VALUES - 1); color[i] = vec3f(alpha[i]); }
    if (!xfFile.empty()) {                                          // 128 raw floats (viewer.cpp:139-145,1147-1152)
      FILE *f = std::fopen(xfFile.c_str(), "rb");
      if (!f || std::fread(alpha.data(), sizeof(float), NUM_XF_VALUES, f) != (size_t)NUM_XF_VALUES) throw std::runtime_error("cannot read " + xfFile);
      std::fclose(f);
    }
    for (size_t c = 0; c < config->scalarFields.size(); c++) {      // viewer.cpp:567-575
      interval<float> dom = config->scalarFields[c]->valueRange;
      if (haveRange) dom = interval<float>(range[0], range[1]);
      renderer.updateXF((int)c, alpha.data(), color, dom, xfScale);
    }
    renderer.updateIsoValues(isoVals, isoChans, isoOn);
    if (!contourPlanes.empty()) {
      // the viewer's contour panel set-up (viewer.cpp:673-690): planes given on the command line are enabled, the
      // others keep their defaults (axis i % 3, offset .5, off); the channels apply when more than one was given
      vec3f n[MAX_CONTOUR_PLANES]; float off[MAX_CONTOUR_PLANES]; int ch[MAX_CONTOUR_PLANES], en[MAX_CONTOUR_PLANES];
      if (contourPlanes.size() / 4 > (size_t)MAX_CONTOUR_PLANES) throw std::runtime_error("too many contour planes");
      for (int i = 0; i < MAX_CONTOUR_PLANES; i++) {
        if (contourPlanes.size() / 4 > (size_t)i) {
          n[i] = vec3f(contourPlanes[4 * i], contourPlanes[4 * i + 1], contourPlanes[4 * i + 2]);
          off[i] = contourPlanes[4 * i + 3]; en[i] = 1;
        } else {
          n[i] = vec3f(i % 3 == 0 ? 1.f : 0.f, i % 3 == 1 ? 1.f : 0.f, i % 3 == 2 ? 1.f : 0.f);
          off[i] = .5f; en[i] = 0;
        }
        ch[i] = (contourChans.size() > 1 && (size_t)i < contourChans.size()) ? contourChans[i] : 0;
      }
      renderer.updateContourPlanes(n, off, ch, en);
    }
    renderer.setSpaceSkipping(skipping);
    renderer.setGradientShadingDVR(gradDVR);
    renderer.setGradientShadingISO(gradISO);
    renderer.frameState.ao.enabled = ao;                            // viewer.cpp:944-959
    renderer.frameState.ao.length = aoLength;
    renderer.frameState.clipBox.enabled = clip;
    if (clip) {
      const box3f wb = renderer.worldSpaceBounds;
      renderer.frameState.clipBox.coords.lower = wb.lower + clipBox.lower * wb.span();
      renderer.frameState.clipBox.coords.upper = wb.lower + clipBox.upper * wb.span();
    }
    if (stats) {       // region statistics as Regions::buildFrom prints them, and the work counters of the first frame
      renderer.updateDt(dt);
      renderer.updateFrameID(0);
      const ExaHipStats s = renderer.renderStats();
      std::printf("regions %zu leafEntries %zu\n", renderer.numRegions, renderer.numLeafEntries);
      std::printf("stats segments %llu samples %llu brick_visits %llu corner_loads %llu nodes_visited %llu\n",
                  (unsigned long long)s.segments, (unsigned long long)s.samples, (unsigned long long)s.brick_visits,
                  (unsigned long long)s.corner_loads, (unsigned long long)s.nodes_visited);
    }
    int accumID = 0;
    double kernelMs = 0;
    const auto t0 = std::chrono::steady_clock::now();
    if (!pipeline) {
      for (int fr = 0; fr < frames; fr++) {                          // viewer.cpp:279-288
        renderer.updateDt(dt);
        renderer.updateFrameID(accumID);
        if (pg) ++accumID;
        renderer.render();
        kernelMs += renderer.stats().kernel_ms;
      }
    } else {
      // two device frames + two pinned host frames: the march of frame k+1 runs while frame k travels to the host
      auto ok = [](hipError_t e, const char *what) { if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e)); };
      ok(hipSetDevice(devices[0]), "hipSetDevice");
      const size_t bytes = fb.size() * sizeof(uint32_t);
      uint32_t *dev[2], *host[2];
      hipStream_t march, copy;
      hipEvent_t rendered[2], copied[2];
      ok(hipStreamCreateWithFlags(&march, hipStreamNonBlocking), "hipStreamCreate");
      ok(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking), "hipStreamCreate");
      for (int k = 0; k < 2; k++) {
        ok(hipMalloc((void **)&dev[k], bytes), "hipMalloc");
        ok(hipHostMalloc((void **)&host[k], bytes, hipHostMallocDefault), "hipHostMalloc");
        ok(hipEventCreateWithFlags(&rendered[k], hipEventDisableTiming), "hipEventCreate");
        ok(hipEventCreateWithFlags(&copied[k], hipEventDisableTiming), "hipEventCreate");
      }
      renderer.updateDt(dt);
      renderer.updateFrameID(0);
      renderer.render();                                   // one synchronous frame first: launch-order feedback
      const auto t1 = std::chrono::steady_clock::now();
      for (int fr = 0; fr < frames; fr++) {
        const int k = fr & 1;
        renderer.updateFrameID(accumID);
        if (pg) ++accumID;
        if (fr >= 2) ok(hipStreamWaitEvent(march, copied[k], 0), "hipStreamWaitEvent");   // buffer k has left the device
        renderer.renderAsync(dev[k], march);
        ok(hipEventRecord(rendered[k], march), "hipEventRecord");
        ok(hipStreamWaitEvent(copy, rendered[k], 0), "hipStreamWaitEvent");
        ok(hipMemcpyAsync(host[k], dev[k], bytes, hipMemcpyDeviceToHost, copy), "hipMemcpyAsync");
        ok(hipEventRecord(copied[k], copy), "hipEventRecord");
      }
      ok(hipStreamSynchronize(copy), "hipStreamSynchronize");
      ok(hipStreamSynchronize(march), "hipStreamSynchronize");
      const double secP = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
      std::memcpy(fb.data(), host[(frames - 1) & 1], bytes);
      kernelMs = renderer.stats().kernel_ms * frames;
      std::printf("pipelined: %d frames in %.3f ms (%.3f ms per frame incl. copy to host)\n", frames, 1000.0 * secP, 1000.0 * secP / frames);
      for (int k = 0; k < 2; k++) { (void)hipFree(dev[k]); (void)hipHostFree(host[k]); (void)hipEventDestroy(rendered[k]); (void)hipEventDestroy(copied[k]); }
      (void)hipStreamDestroy(march); (void)hipStreamDestroy(copy);
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("Avg. after %d frames: %.3f FPS (%.3f ms), kernel %.3f ms\n", frames, frames / sec, 1000.0 * sec / frames, kernelMs / frames);
    if (!outName.empty()) {
      FILE *f = std::fopen(outName.c_str(), "wb");
      if (!f) throw std::runtime_error("cannot write " + outName);
      std::fprintf(f, "P6\n%d %d\n255\n", size.x, size.y);
      for (int y = size.y - 1; y >= 0; y--)
        for (int x = 0; x < size.x; x++) { const uint32_t p = fb[size_t(y) * size.x + x]; const unsigned char rgb[3] = { (unsigned char)(p & 255), (unsigned char)((p >> 8) & 255), (unsigned char)((p >> 16) & 255) }; std::fwrite(rgb, 1, 3, f); }
      std::fclose(f);
    }
    return 0;
  } catch (const std::exception &e) {
    std::cerr << "Fatal error " << e.what() << std::endl;
    return 1;
  }
}
