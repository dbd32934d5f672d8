"""Shared machinery of the parity tests: one `Case` = scene + camera + transfer
function + settings, runnable through the CPU oracle and through the HIP module
(via the C ABI) with identical inputs."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import pyoracle as po  # noqa: E402  (tests are allowed to use the oracle)
from owlexabrick_amd import harness, scenes  # noqa: E402


def ramp_xf(seed=0):
    return harness.default_xf()


def band_xf(lo=0.35, hi=0.65):
    """opacity only inside a value band -> many regions inactive (space skipping)."""
    xf = harness.default_xf()
    t = np.arange(128) / 127.0
    xf[:, 3] = np.where((t >= lo) & (t <= hi), 0.6, 0.0)
    return xf


class Case:
    def __init__(self, scene, W=64, H=64, grad=0, iso=None, xf=None, dt=0.5, opacity_scale=1.0,
                 space_skipping=1, ao=0, ao_length=1e20, clip=None, frameID=0, camera=None,
                 xfm=None, grad_iso=1, multi=True, xf_domains=None, accel=None, fast_math=None, contour=None, meshes=None,
                 tf_filter=None, basis_form=None, allow_empty_cells=False):
        self.scene, self.W, self.H = scene, W, H
        # association of the eight-corner basis sums on BOTH sides (oracle: or_set_basis_form, module: option basis_form):
        # 0 = the reference's source order, 1 = per axis with fused multiply-adds; None = the module's default, which the
        # oracle (whose own default is the source order) is then told to follow (EXA_TEST_BASIS_FORM runs a whole test
        # session in one form)
        env = os.environ.get("EXA_TEST_BASIS_FORM")
        from owlexabrick_amd.binding import DEFAULT_BASIS_FORM
        self.basis_form = basis_form if basis_form is not None else (int(env) if env else DEFAULT_BASIS_FORM)
        self.grad, self.iso, self.dt = grad, iso, dt
        self.allow_empty_cells = allow_empty_cells     # the reference's build option ALLOW_EMPTY_CELLS: cell id -1 = no cell
        if allow_empty_cells:
            self.basis_form = 0                        # an empty cell is a per-corner property: source order only
        self.xfs = xf if isinstance(xf, list) else [xf if xf is not None else ramp_xf()] * len(scene.fields)
        self.opacity_scale, self.space_skipping = opacity_scale, space_skipping
        self.ao, self.ao_length, self.clip, self.frameID = ao, ao_length, clip, frameID
        self.camera, self.xfm, self.grad_iso, self.multi = camera, xfm, grad_iso, multi
        self.accel = accel
        self.fast_math = fast_math
        self.contour = contour
        self.tf_filter = tf_filter    # None = default (CUDA 1.8 fixed-point filter weight); 0 = full-precision weight
        self.meshes = meshes          # list of (verts[n,3], tris[m,3]), world space
        nf = len(scene.fields)
        self.nprim = nf if multi else 1
        self.colormap_channel = 0 if (multi or nf < 2) else 1
        if xf_domains is None:
            xf_domains = []
            for f in scene.fields:
                xf_domains.append((float(min(f.min(), 0.0)), float(max(f.max(), 0.0))))
        self.xf_domains = xf_domains

    def cam(self, lo, hi):
        if isinstance(self.camera, dict):       # explicit pos/dir00/dirDu/dirDv
            return self.camera
        if self.camera is not None:
            return harness.camera(self.camera[0], self.camera[1], self.camera[2], self.camera[3], self.W, self.H)
        if self.xfm is not None:
            # world bounds = inverse-transformed voxel bounds; tests pass world camera explicitly instead
            raise ValueError("xfm cases need an explicit camera")
        return harness.default_camera(lo, hi, self.W, self.H)

    # ---- oracle ----
    def oracle_scene(self):
        S = po.OracleScene(self.scene.bricks7, self.scene.cellIDs, self.scene.fields,
                           num_region_fields=len(self.scene.fields) if self.multi else 1,
                           allow_empty_cells=self.allow_empty_cells)
        for c, xf in enumerate(self.xfs):
            S.set_xf(c, xf)
        if self.tf_filter is not None:
            S.set_tf_filter(self.tf_filter)
        if self.basis_form is not None:
            S.set_basis_form(self.basis_form)
        if self.meshes:
            S.set_triangles(*self._merged_meshes())
        return S

    def _merged_meshes(self):
        verts, tris, base = [], [], 0
        for v, t in self.meshes:
            v = np.asarray(v, dtype=np.float32).reshape(-1, 3)
            verts.append(v)
            tris.append(np.asarray(t, dtype=np.int32).reshape(-1, 3) + base)
            base += len(v)
        return np.concatenate(verts), np.concatenate(tris)

    def oracle_state(self, S, frameID=None):
        lo, hi = S.voxel_bounds()
        fs = po.FrameState()
        harness.fill_frame_state(fs, self.cam(lo, hi), self.xf_domains, xfOpacityScale=self.opacity_scale,
                                 frameID=self.frameID if frameID is None else frameID, iso=self.iso,
                                 clip=self.clip, ao_enabled=self.ao, ao_length=self.ao_length, xfm=self.xfm,
                                 contour=self.contour)
        skipping = int(self.space_skipping and not self.contour)      # OptixRenderer.cpp:418-432
        P = po.Params(self.dt, self.nprim, self.colormap_channel, self.grad, self.grad_iso, self.nprim, skipping)
        return fs, P

    def run_oracle(self, nthreads=8, frames=1, window=None):
        S = self.oracle_scene()
        acc, out = None, None
        for f in range(frames):
            fs, P = self.oracle_state(S, frameID=self.frameID + f)
            rgba, acc, st = S.render(fs, P, self.W, self.H, window=window, accum=acc, nthreads=nthreads)
            out = (rgba, acc, st)
        return out

    # ---- HIP module through the C ABI ----
    def hip_renderer(self, device=0):
        from owlexabrick_amd import binding
        prep = binding.Prep(self.scene, num_region_fields=len(self.scene.fields) if self.multi else 1,
                            allow_empty_cells=self.allow_empty_cells)
        R = binding.Renderer(prep, device=device, multiFieldDvr=self.multi)
        if self.meshes:
            R.setTriangles(*self._merged_meshes())
        if self.accel is not None:
            # 0 = LBVH restart per segment, 1 = region kd-tree with the stack walk, 2 = region kd-tree with the rope walk
            # (None: the module's own choice per frame)
            R.setOption("accel", 1 if self.accel else 0)
            if self.accel:
                R.setOption("walk", self.accel)
        if self.fast_math is not None:
            R.setOption("fast_math", self.fast_math)
        if self.tf_filter is not None:
            R.setOption("tf_filter", self.tf_filter)
        if self.basis_form is not None:
            R.setOption("basis_form", self.basis_form)
        for k, v in getattr(self, "options", {}).items():
            R.setOption(k, v)
        lo, hi = prep.voxel_bounds()
        cam = self.cam(lo, hi)
        if self.xfm is not None:
            R.setVoxelSpaceTransform(self.xfm["vx"], self.xfm["vy"], self.xfm["vz"], self.xfm["p"])
        R.resizeFrameBuffer((self.W, self.H))
        R.updateCamera(cam["pos"], cam["dir00"], cam["dirDu"], cam["dirDv"])
        for c, xf in enumerate(self.xfs):
            R.updateXF(c, xf[:, 3], xf[:, :3], self.xf_domains[c], self.opacity_scale)
        iso_v, iso_c, iso_e = [0.0, 0.0], [0, 0], [0, 0]
        for i, spec in enumerate(self.iso or []):
            iso_v[i], iso_c[i], iso_e[i] = spec[0], spec[1], 1
        R.updateIsoValues(iso_v, iso_c, iso_e)
        if self.contour:
            n = [c[0] for c in self.contour] + [[1, 0, 0]] * (3 - len(self.contour))
            R.updateContourPlanes(n, [c[1] for c in self.contour] + [0.5] * (3 - len(self.contour)),
                                  [c[2] for c in self.contour] + [0] * (3 - len(self.contour)),
                                  [1] * len(self.contour) + [0] * (3 - len(self.contour)))
        R.setSpaceSkipping(bool(self.space_skipping))
        R.setGradientShadingDVR(bool(self.grad))
        R.setGradientShadingISO(bool(self.grad_iso))
        R.updateDt(self.dt)
        R.frameState.ao.enabled, R.frameState.ao.length = int(self.ao), float(self.ao_length)   # viewer.cpp:946-956
        R.frameState.clipBox.enabled = 0
        if self.clip is not None:
            R.frameState.clipBox.enabled = 1
            for i in range(3):
                R.frameState.clipBox.lo[i], R.frameState.clipBox.hi[i] = float(self.clip[0][i]), float(self.clip[1][i])
        return R

    def run_hip(self, frames=1, stats=False):
        R = self.hip_renderer()
        out = None
        for f in range(frames):
            R.updateFrameID(self.frameID + f)       # viewer.cpp:281-288
            if stats and f == frames - 1:
                rgba, st = R.renderStats()
            else:
                rgba, st = R.render(), None
            out = (rgba, R.readAccum(), st)
        R.close()
        return out


# Float tolerance of CPU-vs-GPU parity (stated in DESIGN.md): the two sides differ
# only in libm vs OCML powf/cosf/sinf (<= 2 ulp); everything else is the same
# IEEE operation sequence.
ACCUM_ATOL = 2e-5
ACCUM_RTOL = 1e-4
RGBA_MAX_LSB = 1
# A ray stops when its opacity reaches 0.98 (exabrick.cu:49,1180).  An ulp of difference in a
# transcendental can move that decision by one sample for a rare pixel; the pixel then moves by
# at most the remaining transmittance (0.02) times its colour.  Allowed for at most FLIP_FRACTION
# of the pixels (2 pixels in a frame too small for that to be a whole pixel: 2 of 14 000 random frames had two — one of
# them, seed 803 of the odd-TF-domain family, is pinned in tests/test_gpu_fuzz.py; the floor only ever matters below
# 6 000 pixels, above that FLIP_FRACTION allows three or more anyway); every other pixel must meet ACCUM_ATOL/RTOL.
FLIP_BOUND = 0.021
FLIP_FRACTION = 5e-4


def compare(oracle_out, hip_out, what=""):
    o_rgba, o_acc, _ = oracle_out
    h_rgba, h_acc, _ = hip_out
    da = np.abs(o_acc.astype(np.float64) - h_acc.astype(np.float64))
    tol = ACCUM_ATOL + ACCUM_RTOL * np.abs(o_acc)
    o8 = harness.unpack_rgba8(o_rgba).astype(np.int32)
    h8 = harness.unpack_rgba8(h_rgba).astype(np.int32)
    d8 = np.abs(o8 - h8)
    nflip = int(((da > tol).any(axis=-1)).sum())
    flips_ok = nflip <= max(2, int(FLIP_FRACTION * da.shape[0] * da.shape[1])) and float(da.max()) <= FLIP_BOUND
    return dict(what=what, accum_max=float(da.max()), accum_bad=int((da > tol).sum()), flip_pixels=nflip, flips_ok=flips_ok,
                rgba_max=int(d8.max()), rgba_bad=int((d8 > RGBA_MAX_LSB).sum()),
                rgba_diff_px=int((d8.max(axis=-1) > 0).sum()), exact=bool(np.array_equal(o_acc, h_acc)))
