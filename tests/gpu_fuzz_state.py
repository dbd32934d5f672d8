"""Sweep of seeded random SEQUENCES of state changes on one renderer handle (no oracle involved):
    python tests/gpu_fuzz_state.py FIRST LAST [--keep-going]
A case of tests/fuzz_cases.py, then 6 random setter calls one after the other — camera, transfer function and opacity
scale, iso-surfaces, step, space skipping, gradient-shading switches, clip box, contour planes, AO, frame size, accel —
each followed by two accumulated frames that must equal, bit for bit, the frames of a FRESH renderer created in that state
(exa/OptixRenderer.cpp:418-552: which setter invalidates what; frameID 0 restarts the accumulation)."""
import sys
import time

import numpy as np

from fuzz_cases import _random_xf, random_case


def _frames(R, n=2):
    out = None
    for f in range(n):
        R.updateFrameID(f)
        out = (np.array(R.render(), copy=True), R.readAccum().copy())
    return out


def _same(a, b):
    return a[0].shape == b[0].shape and np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


def check(seed):
    rng = np.random.default_rng(0x57A7E000 + seed)
    case, desc = random_case(seed, grids=bool(rng.uniform() < 0.3))
    case.W, case.H = int(rng.integers(24, 97)), int(rng.integers(16, 65))
    case.xf_domains = [(0.0, 1.0)] * len(case.scene.fields)
    R = case.hip_renderer()
    vlo, vhi = (np.asarray(v, dtype=np.float64) for v in R.voxelSpaceBounds)
    ext = vhi - vlo
    nf = len(case.scene.fields)
    bad, ops = [], []
    for step in range(6):
        op = str(rng.choice(["camera", "xf", "iso", "dt", "skip", "grad", "clip", "contour", "ao", "resize", "accel"]))
        ops.append(op)
        if op == "camera":
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            case.camera = (list(vlo + 0.5 * ext + d * float(rng.uniform(0.2, 2.0)) * ext.max()), list(vlo + rng.uniform(0.2, 0.8, 3) * ext),
                           [0, 1, 0], float(rng.uniform(30, 90)))
            cam = case.cam(vlo, vhi)
            R.updateCamera(cam["pos"], cam["dir00"], cam["dirDu"], cam["dirDv"])
        elif op == "xf":
            c = int(rng.integers(0, nf))
            case.xfs = list(case.xfs)
            case.xfs[c] = _random_xf(rng, str(rng.choice(["ramp", "band", "table", "steps", "faint"])))
            case.opacity_scale = float(rng.choice([1.0, 0.3, 0.05]))
            for k, xf in enumerate(case.xfs):                      # the scale is one value for all channels (viewer.cpp:570-574)
                R.updateXF(k, xf[:, 3], xf[:, :3], case.xf_domains[k], case.opacity_scale)
        elif op == "iso":
            n = int(rng.integers(0, 3))
            case.iso = [(float(rng.uniform(0.1, 0.9)), int(rng.integers(0, nf))) for _ in range(n)] or None
            v, c, e = [0.0, 0.0], [0, 0], [0, 0]
            for i, spec in enumerate(case.iso or []):
                v[i], c[i], e[i] = spec[0], spec[1], 1
            R.updateIsoValues(v, c, e)
        elif op == "dt":
            case.dt = float(rng.choice([0.5, 0.25, 1.0, 0.37, 2.0]))
            R.updateDt(case.dt)
        elif op == "skip":
            case.space_skipping = 1 - case.space_skipping
            R.setSpaceSkipping(bool(case.space_skipping))
        elif op == "grad":
            case.grad, case.grad_iso = int(rng.integers(0, 2)), int(rng.integers(0, 2))
            R.setGradientShadingDVR(bool(case.grad))
            R.setGradientShadingISO(bool(case.grad_iso))
        elif op == "clip":
            if rng.uniform() < 0.4:
                case.clip = None
                R.frameState.clipBox.enabled = 0
            else:
                clo = vlo + rng.uniform(0.0, 0.4, 3) * ext
                case.clip = (list(clo), list(clo + rng.uniform(0.3, 0.6, 3) * ext))
                R.frameState.clipBox.enabled = 1
                for i in range(3):
                    R.frameState.clipBox.lo[i], R.frameState.clipBox.hi[i] = float(case.clip[0][i]), float(case.clip[1][i])
        elif op == "contour":
            n = int(rng.integers(0, 3))
            case.contour = [([float(x) for x in rng.normal(size=3)], float(rng.uniform(0.2, 0.8)), int(rng.integers(0, nf))) for _ in range(n)] or None
            cc = case.contour or []
            R.updateContourPlanes([c[0] for c in cc] + [[1, 0, 0]] * (3 - len(cc)), [c[1] for c in cc] + [0.5] * (3 - len(cc)),
                                  [c[2] for c in cc] + [0] * (3 - len(cc)), [1] * len(cc) + [0] * (3 - len(cc)))
        elif op == "ao":
            case.ao = 1 - int(case.ao)
            case.ao_length = float(rng.choice([1e20, 0.3 * float(ext.max())]))
            R.frameState.ao.enabled, R.frameState.ao.length = int(case.ao), float(case.ao_length)
        elif op == "resize":
            case.W, case.H = int(rng.integers(17, 121)), int(rng.integers(9, 81))
            R.resizeFrameBuffer((case.W, case.H))
            cam = case.cam(vlo, vhi)                                # the viewer re-derives the screen vectors (viewer.cpp:442-450)
            R.updateCamera(cam["pos"], cam["dir00"], cam["dirDu"], cam["dirDv"])
        elif op == "accel":
            case.accel = int(rng.integers(0, 3))           # LBVH restart, kd-tree stack walk, kd-tree rope walk
            R.setOption("accel", 1 if case.accel else 0)
            if case.accel:
                R.setOption("walk", case.accel)
        live = _frames(R)
        F = case.hip_renderer()
        fresh = _frames(F)
        F.close()
        if not _same(live, fresh):
            bad.append(f"after step {step} ({' > '.join(ops)}): live handle differs from a fresh one")
            break
    R.close()
    desc["ops"] = ops
    return bad, desc


if __name__ == "__main__":
    first, last = int(sys.argv[1]), int(sys.argv[2])
    keep = "--keep-going" in sys.argv
    fails, t0 = 0, time.time()
    for seed in range(first, last + 1):
        bad, desc = check(seed)
        if bad:
            fails += 1
            print(f"FAIL seed {seed}: {desc}\n     {bad}", flush=True)
            if not keep:
                break
        elif seed % 10 == 0:
            print(f"seed {seed} ok ({time.time() - t0:.0f}s)", flush=True)
    print(f"{fails} failed of {last - first + 1}, {time.time() - t0:.0f}s", flush=True)
    sys.exit(1 if fails else 0)
