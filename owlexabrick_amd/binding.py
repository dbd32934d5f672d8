"""ctypes binding of the C ABI in include/exa_hip.h (libexa_hip.so) and a
`Renderer` that keeps the method names of the reference's exa::OptixRenderer
(exa/OptixRenderer.h:32-97) so tests read like calls into the reference.

There is no CPU fallback: if the HIP module is missing or no GPU is present,
loading / creating fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# EXA_HIP_LIB: another build of the same module (A/B timing of kernel variants, tools/ab_variants.sh)
LIB_PATH = os.environ.get("EXA_HIP_LIB") or os.path.join(_HERE, "libexa_hip.so")

NUM_XF_VALUES, MAX_CHANNELS, MAX_ISO, MAX_CONTOUR = 128, 10, 2, 3
# the module's default of option "basis_form" (exa_module.cpp): association of the eight-corner basis sums
DEFAULT_BASIS_FORM = 1


class _Iso(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("value", C.c_float), ("channel", C.c_int32)]


class _Contour(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("normal", C.c_float * 3), ("channel", C.c_int32), ("offset", C.c_float)]


class _Clip(C.Structure):
    _fields_ = [("lo", C.c_float * 3), ("hi", C.c_float * 3), ("enabled", C.c_int32)]


class _AO(C.Structure):
    _fields_ = [("length", C.c_float), ("enabled", C.c_int32)]


class ExaHipFrameState(C.Structure):
    _fields_ = [("cam_pos", C.c_float * 3), ("cam_dir00", C.c_float * 3),
                ("cam_dirDu", C.c_float * 3), ("cam_dirDv", C.c_float * 3),
                ("iso", _Iso * MAX_ISO), ("contour", _Contour * MAX_CONTOUR),
                ("clipBox", _Clip), ("ao", _AO), ("clockScale", C.c_float),
                ("xfm_vx", C.c_float * 3), ("xfm_vy", C.c_float * 3),
                ("xfm_vz", C.c_float * 3), ("xfm_p", C.c_float * 3),
                ("frameID", C.c_int32), ("xfDomain", (C.c_float * 2) * MAX_CHANNELS),
                ("xfOpacityScale", C.c_float)]


class ExaHipParams(C.Structure):
    _fields_ = [("dt", C.c_float), ("numPrimaryChannels", C.c_int32), ("colormapChannel", C.c_int32),
                ("gradientShadingDVR", C.c_int32), ("gradientShadingISO", C.c_int32),
                ("numChannels", C.c_int32), ("spaceSkippingEnabled", C.c_int32)]


class ExaHipTracer(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("channels", C.c_int32 * 3), ("numTraces", C.c_int32),
                ("numTimesteps", C.c_int32), ("steplen", C.c_float)]


class ExaHipScene(C.Structure):
    _fields_ = [("bricks", C.c_void_p), ("numBricks", C.c_uint64),
                ("regions", C.c_void_p), ("numRegions", C.c_uint64),
                ("leafList", C.c_void_p), ("leafListSize", C.c_uint64),
                ("scalars", C.c_void_p), ("channelOffset", C.c_void_p),
                ("totalCells", C.c_uint64), ("numFields", C.c_int32),
                ("voxelBounds_lo", C.c_float * 3), ("voxelBounds_hi", C.c_float * 3),
                ("kdNodes", C.c_void_p), ("numKdNodes", C.c_uint64), ("kdRoot", C.c_int32), ("allowEmptyCells", C.c_int32)]


class ExaHipStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("segments", "sample_evals", "samples", "brick_visits", "corner_loads",
                                          "iso_segments", "iso_evals", "nodes_visited", "node_bytes", "pixels")] + \
               [("diag", C.c_uint64 * 9), ("phase_cycles", C.c_uint64 * 5), ("kernel_ms", C.c_float), ("rebuild_ms", C.c_float),
                ("walk_restarts", C.c_uint64), ("walk_union_nodes", C.c_uint64), ("walk_probe_overflow", C.c_uint64),
                ("wave_iters", C.c_uint64), ("tile_iters", C.c_uint64), ("walk_leaf_visits", C.c_uint64)]

    def asdict(self):
        d = {}
        for n, t in self._fields_:
            v = getattr(self, n)
            d[n] = int(v) if t is C.c_uint64 else (float(v) if t is C.c_float else [int(x) for x in v])
        return d


BRICK_DTYPE = np.dtype([("lower", "<i4", 3), ("size", "<i4", 3), ("level", "<i4"), ("begin", "<u4")])
REGION_DTYPE = np.dtype([("dom_lo", "<f4", 3), ("dom_hi", "<f4", 3), ("vr_lo", "<f4"), ("vr_hi", "<f4"),
                         ("leafListBegin", "<i4"), ("leafListSize", "<i4"), ("finestLevelCellWidth", "<f4")])
KDNODE_DTYPE = np.dtype([("split", "<f4"), ("axis", "<i4"), ("left", "<i4"), ("right", "<i4")])
KD_EMPTY = -2 ** 31

# every symbol include/exa_hip.h declares
ABI_SYMBOLS = ["exa_prep_create", "exa_prep_create_ex", "exa_prep_destroy", "exa_prep_scene", "exa_prep_last_error", "exa_prep_ropes",
               "exa_hip_create", "exa_hip_create_multi", "exa_hip_destroy", "exa_hip_resize", "exa_hip_set_frame_state",
               "exa_hip_set_xf", "exa_hip_set_triangles", "exa_hip_reset_tracer", "exa_hip_set_tracer_enabled",
               "exa_hip_advance_tracer", "exa_hip_read_traces", "exa_hip_set_params", "exa_hip_set_shard", "exa_hip_output_pixels",
               "exa_hip_untile", "exa_hip_render", "exa_hip_render_stats", "exa_hip_get_stats",
               "exa_hip_read_accum", "exa_hip_write_accum", "exa_hip_read_activity",
               "exa_hip_set_option", "exa_hip_last_error"]

_lib = None


def lib():
    """load libexa_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            # not built yet: try to build the HIP module (hipcc cross-compiles without a GPU); never a CPU fallback
            import subprocess
            try:
                subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-s", "-j6"])
            except Exception as e:  # noqa: BLE001
                raise RuntimeError(f"{LIB_PATH} not found and building it failed ({e}); run "
                                   "__graft_entry__.build(); there is no CPU fallback") from e
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.exa_prep_create.restype = C.c_int
        L.exa_prep_create.argtypes = [vp, C.c_uint64, vp, C.c_uint64, C.POINTER(vp), C.POINTER(C.c_uint64),
                                      C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
        L.exa_prep_create_ex.restype = C.c_int
        L.exa_prep_create_ex.argtypes = [vp, C.c_uint64, vp, C.c_uint64, C.POINTER(vp), C.POINTER(C.c_uint64),
                                         C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
        L.exa_prep_destroy.argtypes = [vp]
        L.exa_prep_ropes.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), vp, vp, vp, vp, C.POINTER(C.c_int32)]
        L.exa_prep_scene.argtypes = [vp, C.POINTER(ExaHipScene)]
        L.exa_prep_last_error.restype = C.c_char_p
        L.exa_hip_create.argtypes = [C.POINTER(ExaHipScene), C.c_int32, C.POINTER(vp)]
        L.exa_hip_create_multi.argtypes = [C.POINTER(ExaHipScene), C.POINTER(C.c_int32), C.c_int32, C.POINTER(vp)]
        L.exa_hip_destroy.argtypes = [vp]
        L.exa_hip_resize.argtypes = [vp, C.c_int32, C.c_int32]
        L.exa_hip_set_frame_state.argtypes = [vp, C.POINTER(ExaHipFrameState)]
        L.exa_hip_set_xf.argtypes = [vp, C.c_int32, vp]
        L.exa_hip_set_params.argtypes = [vp, C.POINTER(ExaHipParams)]
        L.exa_hip_set_triangles.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64]
        L.exa_hip_reset_tracer.argtypes = [vp, C.POINTER(ExaHipTracer), vp]
        L.exa_hip_set_tracer_enabled.argtypes = [vp, C.c_int32]
        L.exa_hip_advance_tracer.argtypes = [vp, C.POINTER(C.c_int32)]
        L.exa_hip_read_traces.argtypes = [vp, vp]
        L.exa_hip_set_shard.argtypes = [vp, C.c_int32, C.c_int32]
        L.exa_hip_output_pixels.restype = C.c_uint64
        L.exa_hip_output_pixels.argtypes = [vp]
        L.exa_hip_untile.argtypes = [vp, vp, C.c_uint64, C.c_int32, vp, vp]
        L.exa_hip_render.argtypes = [vp, vp, C.c_int32, vp, C.c_int32]
        L.exa_hip_render_stats.argtypes = [vp, vp, C.c_int32, C.POINTER(ExaHipStats)]
        L.exa_hip_get_stats.argtypes = [vp, C.POINTER(ExaHipStats)]
        L.exa_hip_read_accum.argtypes = [vp, vp]
        L.exa_hip_write_accum.argtypes = [vp, vp]
        L.exa_hip_read_activity.argtypes = [vp, C.c_int32, vp]
        L.exa_hip_set_option.argtypes = [vp, C.c_char_p, C.c_int32]
        L.exa_hip_last_error.restype = C.c_char_p
        L.exa_hip_last_error.argtypes = [vp]
        _lib = L
    return _lib


class Prep:
    """host data preparation of the OptixRenderer constructor (exa_prep_*)."""

    def __init__(self, scene, num_region_fields=None, num_threads=0, allow_empty_cells=False):
        """allow_empty_cells: the reference's build option ALLOW_EMPTY_CELLS (cell id -1 = no cell, EXA_PREP_ALLOW_EMPTY_CELLS)"""
        L = lib()
        self.bricks7 = np.ascontiguousarray(scene.bricks7, dtype=np.int32).reshape(-1, 7)
        self.cellIDs = np.ascontiguousarray(scene.cellIDs, dtype=np.int32)
        self.fields = [np.ascontiguousarray(f, dtype=np.float32) for f in scene.fields]
        nf = len(self.fields)
        ptrs = (C.c_void_p * max(nf, 1))(*[f.ctypes.data for f in self.fields])
        lens = (C.c_uint64 * max(nf, 1))(*[f.size for f in self.fields])
        self.h = C.c_void_p()
        rc = L.exa_prep_create_ex(self.bricks7.ctypes.data, self.bricks7.shape[0], self.cellIDs.ctypes.data,
                                  self.cellIDs.size, ptrs, lens, nf,
                                  nf if num_region_fields is None else num_region_fields, num_threads,
                                  1 if allow_empty_cells else 0, C.byref(self.h))
        if rc:
            raise RuntimeError(L.exa_prep_last_error().decode())
        self.scene = ExaHipScene()
        L.exa_prep_scene(self.h, C.byref(self.scene))
        self.num_fields = nf

    def _arr(self, ptr, count, dtype):
        if count == 0:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype)

    def bricks(self):
        return self._arr(self.scene.bricks, self.scene.numBricks, BRICK_DTYPE)

    def regions(self):
        return self._arr(self.scene.regions, self.scene.numRegions, REGION_DTYPE)

    def leaflist(self):
        return self._arr(self.scene.leafList, self.scene.leafListSize, np.int32)

    def scalars(self):
        return self._arr(self.scene.scalars, self.scene.numFields * self.scene.totalCells, np.float32)

    def kd_nodes(self):
        return self._arr(self.scene.kdNodes, self.scene.numKdNodes, KDNODE_DTYPE)

    def ropes(self):
        """leaves and neighbour links of the rope walk as the module builds them (exa_prep_ropes, a diagnostic):
        dict(boxes [n,6], links [n,6], region [n], nodes (KDNODE_DTYPE), flags)"""
        nl, nn, flags = C.c_uint64(0), C.c_uint64(0), C.c_int32(0)
        if lib().exa_prep_ropes(self.h, C.byref(nl), C.byref(nn), None, None, None, None, None):
            raise RuntimeError(lib().exa_prep_last_error().decode())
        boxes = np.zeros((nl.value, 6), dtype=np.float32)
        links = np.zeros((nl.value, 6), dtype=np.int32)
        region = np.zeros(nl.value, dtype=np.int32)
        nodes = np.zeros(max(1, nn.value), dtype=KDNODE_DTYPE)
        if lib().exa_prep_ropes(self.h, C.byref(nl), C.byref(nn), boxes.ctypes.data, links.ctypes.data, region.ctypes.data,
                                nodes.ctypes.data, C.byref(flags)):
            raise RuntimeError(lib().exa_prep_last_error().decode())
        return dict(boxes=boxes, links=links, region=region, nodes=nodes[:nn.value], flags=int(flags.value))

    def voxel_bounds(self):
        return (np.array(self.scene.voxelBounds_lo, dtype=np.float32),
                np.array(self.scene.voxelBounds_hi, dtype=np.float32))

    def close(self):
        if getattr(self, "h", None):
            lib().exa_prep_destroy(self.h)
            self.h = None

    __del__ = close


class Renderer:
    """Python mirror of exa::OptixRenderer's public methods over the C ABI."""

    def __init__(self, prep, device=0, multiFieldDvr=True, devices=None):
        """devices: a list of device indices -> one handle that drives all of them (exa_hip_create_multi)"""
        L = lib()
        self.prep = prep
        self.h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int32 * len(devices))(*devices)
            rc = L.exa_hip_create_multi(C.byref(prep.scene), arr, len(devices), C.byref(self.h))
        else:
            rc = L.exa_hip_create(C.byref(prep.scene), device, C.byref(self.h))
        if rc:
            raise RuntimeError(L.exa_hip_last_error(None).decode())
        self.numFields = prep.num_fields
        self.frameState = ExaHipFrameState()
        self.frameState.xfm_vx[0] = self.frameState.xfm_vy[1] = self.frameState.xfm_vz[2] = 1.0
        self.frameState.ao.length, self.frameState.ao.enabled = 1e20, 1   # FrameState.h:55-58 defaults
        self.frameState.xfOpacityScale = 1.0
        for i in range(MAX_CONTOUR):
            self.frameState.contour[i].normal[0] = 1.0
            self.frameState.contour[i].offset = 0.5
        nprim = self.numFields if multiFieldDvr else 1
        self.params = ExaHipParams(0.5, nprim, 0 if (multiFieldDvr or self.numFields < 2) else 1, 1, 1, nprim, 1)
        self.doSpaceSkipping = True
        self.fbSize = (0, 0)
        self.voxelSpaceBounds = prep.voxel_bounds()

    def _check(self, rc):
        if rc:
            raise RuntimeError(lib().exa_hip_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            lib().exa_hip_destroy(self.h)
            self.h = None

    __del__ = close

    # ---- OptixRenderer method set (exa/OptixRenderer.h:38-79) ----
    def setVoxelSpaceTransform(self, vx, vy, vz, p):
        for i in range(3):
            self.frameState.xfm_vx[i], self.frameState.xfm_vy[i] = float(vx[i]), float(vy[i])
            self.frameState.xfm_vz[i], self.frameState.xfm_p[i] = float(vz[i]), float(p[i])

    def resizeFrameBuffer(self, fbSize):
        self.fbSize = (int(fbSize[0]), int(fbSize[1]))
        self._check(lib().exa_hip_resize(self.h, *self.fbSize))

    def updateIsoValues(self, isoValues, channels, enabled):
        for i in range(MAX_ISO):
            self.frameState.iso[i].value = float(isoValues[i])
            self.frameState.iso[i].channel = int(channels[i])
            self.frameState.iso[i].enabled = int(enabled[i])

    # ---- streamline tracer (OptixRenderer::setTracerEnabled / resetTracer / advanceTracer) ----
    def resetTracer(self, seeds, channels=(0, 1, 2), numTimesteps=100, steplen=1e-6, enabled=True):
        sd = np.ascontiguousarray(seeds, dtype=np.float32).reshape(-1, 3)
        self._tracer = ExaHipTracer(int(enabled), (C.c_int32 * 3)(*channels), sd.shape[0], int(numTimesteps), float(steplen))
        self._check(lib().exa_hip_reset_tracer(self.h, C.byref(self._tracer), sd.ctypes.data))

    def setTracerEnabled(self, enable):
        self._check(lib().exa_hip_set_tracer_enabled(self.h, int(bool(enable))))

    def advanceTracer(self):
        r = C.c_int32(0)
        self._check(lib().exa_hip_advance_tracer(self.h, C.byref(r)))
        return bool(r.value)

    def readTraces(self):
        out = np.zeros((self._tracer.numTraces, self._tracer.numTimesteps, 3), dtype=np.float32)
        self._check(lib().exa_hip_read_traces(self.h, out.ctypes.data))
        return out

    def setTriangles(self, verts, tris):
        """the `surfaces` argument of the OptixRenderer constructor, all meshes concatenated"""
        v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
        self._check(lib().exa_hip_set_triangles(self.h, v.ctypes.data, v.shape[0], t.ctypes.data, t.shape[0]))

    def updateContourPlanes(self, normals, offsets, channels, enabled):
        for i in range(MAX_CONTOUR):
            n = np.asarray(normals[i], dtype=np.float32)
            n = (n * (np.float32(1.0) / np.sqrt(np.dot(n, n), dtype=np.float32))).astype(np.float32)   # normalize() (:511)
            for k in range(3):
                self.frameState.contour[i].normal[k] = float(n[k])
            self.frameState.contour[i].offset = float(offsets[i])
            self.frameState.contour[i].channel = int(channels[i])
            self.frameState.contour[i].enabled = int(enabled[i])

    def updateCamera(self, pos, dir00, dirDu, dirDv):
        for i in range(3):
            self.frameState.cam_pos[i], self.frameState.cam_dir00[i] = float(pos[i]), float(dir00[i])
            self.frameState.cam_dirDu[i], self.frameState.cam_dirDv[i] = float(dirDu[i]), float(dirDv[i])

    def updateXF(self, chan, opacities, colorMap, xfDomain, xfOpacityScale=0.1):
        colorMap = np.asarray(colorMap, dtype=np.float32)
        if colorMap.shape[0] != NUM_XF_VALUES:
            raise RuntimeError("mismatching xf size!?")          # OptixRenderer.cpp:382-383
        lut = np.concatenate([colorMap[:, :3], np.asarray(opacities, dtype=np.float32).reshape(-1, 1)], axis=1)
        lut = np.ascontiguousarray(lut, dtype=np.float32)
        self.frameState.xfDomain[chan][0], self.frameState.xfDomain[chan][1] = float(xfDomain[0]), float(xfDomain[1])
        self.frameState.xfOpacityScale = float(xfOpacityScale)
        self._check(lib().exa_hip_set_xf(self.h, chan, lut.ctypes.data))

    def updateFrameID(self, frameID):
        self.frameState.frameID = int(frameID)

    def updateDt(self, dt):
        self.params.dt = float(dt)

    def setSpaceSkipping(self, enable):
        self.doSpaceSkipping = bool(enable)

    def setGradientShadingDVR(self, enable):
        self.params.gradientShadingDVR = int(bool(enable))

    def setGradientShadingISO(self, enable):
        self.params.gradientShadingISO = int(bool(enable))

    def setShard(self, rank, world):
        self._check(lib().exa_hip_set_shard(self.h, rank, world))

    def setOption(self, key, value):
        self._check(lib().exa_hip_set_option(self.h, key.encode(), int(value)))

    def _push_state(self):
        contour = any(self.frameState.contour[i].enabled for i in range(MAX_CONTOUR))
        self.params.spaceSkippingEnabled = int((not contour) and self.doSpaceSkipping)  # OptixRenderer.cpp:418-432
        self._check(lib().exa_hip_set_frame_state(self.h, C.byref(self.frameState)))
        self._check(lib().exa_hip_set_params(self.h, C.byref(self.params)))

    def outputPixels(self):
        return int(lib().exa_hip_output_pixels(self.h))

    def render(self, device_ptr=None, stream=None, async_=False):
        """OptixRenderer::render().  Without device_ptr returns the colour buffer as a
        numpy array ([H,W] uint32 for a whole frame, flat tile-major for a shard)."""
        self._push_state()
        if device_ptr is not None:
            self._check(lib().exa_hip_render(self.h, C.c_void_p(device_ptr), 1, C.c_void_p(stream or 0), int(async_)))
            return None
        n = self.outputPixels()
        out = np.zeros(n, dtype=np.uint32)
        self._check(lib().exa_hip_render(self.h, out.ctypes.data, 0, None, 0))
        W, H = self.fbSize
        return out.reshape(H, W) if n == W * H else out

    def renderStats(self):
        self._push_state()
        n = self.outputPixels()
        out = np.zeros(n, dtype=np.uint32)
        st = ExaHipStats()
        self._check(lib().exa_hip_render_stats(self.h, out.ctypes.data, 0, C.byref(st)))
        W, H = self.fbSize
        return (out.reshape(H, W) if n == W * H else out), st.asdict()

    def stats(self):
        st = ExaHipStats()
        self._check(lib().exa_hip_get_stats(self.h, C.byref(st)))
        return st.asdict()

    def readAccum(self):
        n = self.outputPixels()
        out = np.zeros((n, 4), dtype=np.float32)
        self._check(lib().exa_hip_read_accum(self.h, out.ctypes.data))
        W, H = self.fbSize
        return out.reshape(H, W, 4) if n == W * H else out

    def writeAccum(self, accum):
        a = np.ascontiguousarray(accum, dtype=np.float32)
        self._check(lib().exa_hip_write_accum(self.h, a.ctypes.data))

    def readActivity(self, which=0):
        self._push_state()
        out = np.zeros(self.prep.scene.numRegions, dtype=np.uint8)
        self._check(lib().exa_hip_read_activity(self.h, which, out.ctypes.data))
        return out

    def untile(self, gathered_ptr, shard_stride, world, out_ptr, stream=None):
        self._check(lib().exa_hip_untile(self.h, C.c_void_p(gathered_ptr), shard_stride, world,
                                         C.c_void_p(out_ptr), C.c_void_p(stream or 0)))
