"""ctypes binding of the CPU oracle (oracle/libexa_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under owlexabrick_amd/ imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libexa_oracle.so")

NUM_XF_VALUES, MAX_CHANNELS, MAX_ISO, MAX_CONTOUR = 128, 10, 2, 3


class _Iso(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("value", C.c_float), ("channel", C.c_int32)]


class _Contour(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("normal", C.c_float * 3), ("channel", C.c_int32), ("offset", C.c_float)]


class _Clip(C.Structure):
    _fields_ = [("lo", C.c_float * 3), ("hi", C.c_float * 3), ("enabled", C.c_int32)]


class _AO(C.Structure):
    _fields_ = [("length", C.c_float), ("enabled", C.c_int32)]


class FrameState(C.Structure):
    """programs/FrameState.h:29-71 (texture handles replaced by set_xf)."""
    _fields_ = [("cam_pos", C.c_float * 3), ("cam_dir00", C.c_float * 3),
                ("cam_dirDu", C.c_float * 3), ("cam_dirDv", C.c_float * 3),
                ("iso", _Iso * MAX_ISO), ("contour", _Contour * MAX_CONTOUR),
                ("clipBox", _Clip), ("ao", _AO), ("clockScale", C.c_float),
                ("xfm_vx", C.c_float * 3), ("xfm_vy", C.c_float * 3),
                ("xfm_vz", C.c_float * 3), ("xfm_p", C.c_float * 3),
                ("frameID", C.c_int32), ("xfDomain", (C.c_float * 2) * MAX_CHANNELS),
                ("xfOpacityScale", C.c_float)]


class Params(C.Structure):
    _fields_ = [("dt", C.c_float), ("numPrimaryChannels", C.c_int32), ("colormapChannel", C.c_int32),
                ("gradientShadingDVR", C.c_int32), ("gradientShadingISO", C.c_int32),
                ("numChannels", C.c_int32), ("spaceSkippingEnabled", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("segments", "sample_evals", "samples", "brick_visits",
                                          "corner_loads", "iso_segments", "iso_evals")]

    def asdict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


BRICK_DTYPE = np.dtype([("lower", "<i4", 3), ("size", "<i4", 3), ("level", "<i4"), ("begin", "<u4")])
REGION_DTYPE = np.dtype([("dom_lo", "<f4", 3), ("dom_hi", "<f4", 3), ("vr_lo", "<f4"), ("vr_hi", "<f4"),
                         ("leafListBegin", "<i4"), ("leafListSize", "<i4"), ("finestLevelCellWidth", "<f4")])
assert BRICK_DTYPE.itemsize == 32 and REGION_DTYPE.itemsize == 44


def build(force=False):
    """compile the C restatement (gcc); building the checker is not using it."""
    src = [os.path.join(_HERE, f) for f in ("exa_oracle.c", "exa_oracle.h", "Makefile")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        vp, sz, fp, ip = C.c_void_p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.or_scene_create.restype = vp
        L.or_scene_create.argtypes = [vp, sz, vp, sz, C.POINTER(vp), C.POINTER(sz), C.c_int, C.c_int, C.c_char_p, sz]
        L.or_scene_create_ex.restype = vp
        L.or_scene_create_ex.argtypes = [vp, sz, vp, sz, C.POINTER(vp), C.POINTER(sz), C.c_int, C.c_int, C.c_int, C.c_char_p, sz]
        L.or_scene_destroy.argtypes = [vp]
        for n in ("or_num_bricks", "or_num_regions", "or_leaflist_size", "or_total_cells"):
            getattr(L, n).restype = sz
            getattr(L, n).argtypes = [vp]
        for n in ("or_bricks", "or_regions", "or_leaflist", "or_scalars"):
            getattr(L, n).restype = vp
            getattr(L, n).argtypes = [vp]
        L.or_voxel_bounds.argtypes = [vp, fp, fp]
        L.or_set_xf.argtypes = [vp, C.c_int, vp]
        L.or_set_tf_filter.argtypes = [vp, C.c_int]
        L.or_set_basis_form.argtypes = [vp, C.c_int]
        L.or_set_triangles.argtypes = [vp, vp, sz, vp, sz]
        L.or_reset_tracer.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_float, vp]
        L.or_advance_tracer.restype = C.c_int
        L.or_advance_tracer.argtypes = [vp]
        L.or_traces.restype = vp
        L.or_traces.argtypes = [vp]
        L.or_tracer_timestep.restype = C.c_int
        L.or_tracer_timestep.argtypes = [vp]
        L.or_volume_active.argtypes = [vp, C.POINTER(FrameState), C.POINTER(Params), vp]
        L.or_iso_active.argtypes = [vp, C.POINTER(FrameState), vp]
        L.or_render.argtypes = [vp, C.POINTER(FrameState), C.POINTER(Params)] + [C.c_int] * 6 + [vp, vp, C.POINTER(Stats), C.c_int]
        L.or_lcg_init_next.restype = C.c_float
        L.or_lcg_init_next.argtypes = [C.c_uint32, C.c_uint32, C.c_int, vp]
        L.or_linear_to_srgb.restype = C.c_float
        L.or_linear_to_srgb.argtypes = [C.c_float]
        L.or_make_8bit.restype = C.c_int32
        L.or_make_8bit.argtypes = [C.c_float]
        L.or_make_rgba8.restype = C.c_uint32
        L.or_make_rgba8.argtypes = [C.c_float] * 3
        L.or_lookup_xf.argtypes = [vp, C.POINTER(FrameState), C.c_float, C.c_int, fp]
        L.or_box_test.restype = C.c_int
        L.or_box_test.argtypes = [fp, fp, C.c_float, C.c_float, fp, fp, fp, fp]
        L.or_sample_point.restype = C.c_int
        L.or_sample_point.argtypes = [vp, C.c_int, fp, C.c_int, C.c_int, fp, fp]
        L.or_trace_region.restype = C.c_int
        L.or_trace_region.argtypes = [vp, vp, fp, fp, C.c_float, C.c_float, fp, fp]
        _lib = L
    return _lib


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class OracleScene:
    """OptixRenderer ctor data prep + ExaBrickRegions::buildFrom, restated on the CPU."""

    def __init__(self, bricks7, cellIDs, fields, num_region_fields=None, allow_empty_cells=False):
        L = lib()
        self.bricks7 = np.ascontiguousarray(bricks7, dtype=np.int32).reshape(-1, 7)
        self.cellIDs = np.ascontiguousarray(cellIDs, dtype=np.int32)
        self.fields = [np.ascontiguousarray(f, dtype=np.float32) for f in fields]
        nf = len(self.fields)
        if num_region_fields is None:
            num_region_fields = nf
        ptrs = (C.c_void_p * max(nf, 1))(*[f.ctypes.data for f in self.fields])
        lens = (C.c_size_t * max(nf, 1))(*[f.size for f in self.fields])
        err = C.create_string_buffer(256)
        self.h = L.or_scene_create_ex(self.bricks7.ctypes.data, self.bricks7.shape[0],
                                      self.cellIDs.ctypes.data, self.cellIDs.size,
                                      ptrs, lens, nf, num_region_fields, int(bool(allow_empty_cells)), err, 256)
        if not self.h:
            raise RuntimeError(err.value.decode())
        self.num_fields = nf

    def close(self):
        if getattr(self, "h", None):
            lib().or_scene_destroy(self.h)
            self.h = None

    __del__ = close

    def _arr(self, fn, count, dtype):
        p = getattr(lib(), fn)(self.h)
        if count == 0:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=dtype).copy()

    @property
    def num_regions(self):
        return lib().or_num_regions(self.h)

    @property
    def total_cells(self):
        return lib().or_total_cells(self.h)

    def bricks(self):
        return self._arr("or_bricks", lib().or_num_bricks(self.h), BRICK_DTYPE)

    def regions(self):
        return self._arr("or_regions", self.num_regions, REGION_DTYPE)

    def leaflist(self):
        return self._arr("or_leaflist", lib().or_leaflist_size(self.h), np.int32)

    def scalars(self):
        return self._arr("or_scalars", self.num_fields * self.total_cells, np.float32)

    def voxel_bounds(self):
        lo, hi = (C.c_float * 3)(), (C.c_float * 3)()
        lib().or_voxel_bounds(self.h, lo, hi)
        return np.array(lo, dtype=np.float32), np.array(hi, dtype=np.float32)

    def set_xf(self, chan, rgba128):
        a = np.ascontiguousarray(rgba128, dtype=np.float32).reshape(128, 4)
        lib().or_set_xf(self.h, chan, a.ctypes.data)

    def set_tf_filter(self, cuda_fixed_point):
        """1 (default) = CUDA tex1D filter weight in 1.8 fixed point, 0 = full precision"""
        lib().or_set_tf_filter(self.h, int(cuda_fixed_point))

    def set_basis_form(self, form):
        """0 (default) = addBasisFunctions in the reference's source order, 1 = the per-axis association with fmaf"""
        lib().or_set_basis_form(self.h, int(form))

    def reset_tracer(self, enabled, channels, num_traces, num_timesteps, steplen, seeds):
        sd = np.ascontiguousarray(seeds, dtype=np.float32).reshape(num_traces, 3)
        self._tracer = (num_traces, num_timesteps)
        lib().or_reset_tracer(self.h, int(enabled), (C.c_int * 3)(*channels), num_traces, num_timesteps, steplen, sd.ctypes.data)

    def advance_tracer(self):
        return bool(lib().or_advance_tracer(self.h))

    def traces(self):
        n, nt = self._tracer
        buf = (C.c_char * (n * nt * 12)).from_address(lib().or_traces(self.h))
        return np.frombuffer(buf, dtype=np.float32).reshape(n, nt, 3).copy()

    def set_triangles(self, verts, tris):
        v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
        lib().or_set_triangles(self.h, v.ctypes.data, v.shape[0], t.ctypes.data, t.shape[0])

    def volume_active(self, fs, params):
        out = np.zeros(self.num_regions, dtype=np.uint8)
        lib().or_volume_active(self.h, C.byref(fs), C.byref(params), out.ctypes.data)
        return out

    def iso_active(self, fs):
        out = np.zeros(self.num_regions, dtype=np.uint8)
        lib().or_iso_active(self.h, C.byref(fs), out.ctypes.data)
        return out

    def render(self, fs, params, W, H, window=None, accum=None, nthreads=1):
        """returns (rgba[H,W] uint32, accum[H,W,4] float32, stats dict); row 0 = bottom."""
        x0, y0, x1, y1 = window if window is not None else (0, 0, W, H)
        rgba = np.zeros((H, W), dtype=np.uint32)
        if accum is None:
            accum = np.zeros((H, W, 4), dtype=np.float32)
        else:
            accum = np.ascontiguousarray(accum, dtype=np.float32).copy()
        st = Stats()
        lib().or_render(self.h, C.byref(fs), C.byref(params), W, H, x0, y0, x1, y1,
                        rgba.ctypes.data, accum.ctypes.data, C.byref(st), nthreads)
        return rgba, accum, st.asdict()

    def sample_point(self, region, pos, chan=0, with_derivative=False):
        v = C.c_float(0)
        g = (C.c_float * 3)()
        ok = lib().or_sample_point(self.h, int(region), _f3(pos), chan, int(with_derivative), C.byref(v), g)
        return bool(ok), np.float32(v.value), np.array(g, dtype=np.float32)

    def lookup_xf(self, fs, v, chan=0):
        out = (C.c_float * 4)()
        lib().or_lookup_xf(self.h, C.byref(fs), float(v), chan, out)
        return np.array(out, dtype=np.float32)

    def trace_region(self, active, org, dir, tmin, tmax):
        t0, t1 = C.c_float(0), C.c_float(0)
        a = np.ascontiguousarray(active, dtype=np.uint8)
        r = lib().or_trace_region(self.h, a.ctypes.data, _f3(org), _f3(dir), tmin, tmax, C.byref(t0), C.byref(t1))
        return r, np.float32(t0.value), np.float32(t1.value)


def lcg(seed0, seed1, n):
    out = np.zeros(n, dtype=np.float32)
    lib().or_lcg_init_next(seed0 & 0xFFFFFFFF, seed1 & 0xFFFFFFFF, n, out.ctypes.data)
    return out


def box_test(org, dir, tmin, tmax, lo, hi):
    t0, t1 = C.c_float(0), C.c_float(0)
    hit = lib().or_box_test(_f3(org), _f3(dir), tmin, tmax, _f3(lo), _f3(hi), C.byref(t0), C.byref(t1))
    return bool(hit), np.float32(t0.value), np.float32(t1.value)
