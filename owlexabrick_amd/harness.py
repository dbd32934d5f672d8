"""Headless stand-in for exa/viewer.cpp's call sequence into the renderer:
camera set-up, default transfer function and default render settings.
Pure numpy float32; produces plain dict/array inputs for any backend."""
import math

import numpy as np

f32 = np.float32


def _normalize(v):
    v = np.asarray(v, dtype=f32)
    return (v * (f32(1.0) / np.sqrt(np.dot(v, v), dtype=f32))).astype(f32)


def _cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=f32)


def camera(origin, interest, up, fovy_deg, W, H):
    """glutViewer/Camera.cpp:94-120 (setOrientation + forceUpFrame) followed by
    glutViewer/OWLViewer.cpp:81-109 (SimpleCamera) and exa/viewer.cpp:226-238
    (cameraChanged): returns pos, dir00, dirDu, dirDv for updateCamera()."""
    origin, interest, up = (np.asarray(x, dtype=f32) for x in (origin, interest, up))
    if np.all(interest == origin):
        vz = np.array([0, 0, 1], dtype=f32)
    else:
        vz = (-_normalize(interest - origin)).astype(f32)
    vx = _cross(up, vz)
    vx = np.array([0, 1, 0], dtype=f32) if np.dot(vx, vx) < 1e-8 else _normalize(vx)
    vy = _normalize(_cross(vz, vx))
    focal = f32(np.sqrt(np.dot(interest - origin, interest - origin), dtype=f32))
    if abs(float(np.dot(vz, up))) >= 1e-6:  # forceUpFrame
        vx = _normalize(_cross(up, vz))
        vy = _normalize(_cross(vz, vx))
    eps = lambda v: max(abs(float(c)) for c in v) * (1.0 / (1 << 21))
    min_focal = f32(max(eps(origin), eps(vx)))
    fd = max(min_focal, focal)
    screen_height = f32(f32(2.0) * f32(math.tan(float(f32(fovy_deg) / f32(2.0) * f32(math.pi) / f32(180.0)))) * fd)
    aspect = f32(W / float(H))
    vertical = (screen_height * vy).astype(f32)
    horizontal = (screen_height * aspect * vx).astype(f32)
    lower_left = (-fd * vz - f32(0.5) * vertical - f32(0.5) * horizontal).astype(f32)
    return dict(pos=origin.astype(f32), dir00=lower_left,
                dirDu=(horizontal / f32(W)).astype(f32), dirDv=(vertical / f32(H)).astype(f32))


def default_camera(bounds_lo, bounds_hi, W, H, fovy=70.0):
    """exa/viewer.cpp:1289-1294: from = center + (-.3,.7,1)*span, at = center, up = +y."""
    lo, hi = np.asarray(bounds_lo, dtype=f32), np.asarray(bounds_hi, dtype=f32)
    center = (f32(0.5) * (lo + hi)).astype(f32)
    span = (hi - lo).astype(f32)
    origin = (center + np.array([-.3, .7, 1.0], dtype=f32) * span).astype(f32)
    return camera(origin, center, [0, 1, 0], fovy, W, H)


def closeup_camera(bounds_lo, bounds_hi, W, H, fovy=70.0):
    """SURVEY 8(d)'s second camera: inside the volume, ahead of and beside the feature at the centre, looking back
    across it along the long axis — every ray starts inside the grid and most cross the refined zone lengthways
    (most regions per ray).  (For the exajet-like scene: off the nose, looking over the wing into the wake.)"""
    lo, hi = np.asarray(bounds_lo, dtype=f32), np.asarray(bounds_hi, dtype=f32)
    center = (f32(0.5) * (lo + hi)).astype(f32)
    span = (hi - lo).astype(f32)
    origin = (center + np.array([-.14, .10, .16], dtype=f32) * span).astype(f32)
    at = (center + np.array([.15, .0, .0], dtype=f32) * span).astype(f32)
    return camera(origin, at, [0, 1, 0], fovy, W, H)


def default_xf(n=128):
    """exa/viewer.cpp:557-565: alpha ramp i/(n-1).  RGB stands in for the embedded
    cool-warm colormap PNG (exa/ColorMapper is a UI asset, out of scope): an
    analytic blue->white->red diverging ramp."""
    t = (np.arange(n, dtype=f32) / f32(n - 1)).astype(f32)
    r = np.clip(0.23 + 1.54 * t, 0, 1) * np.where(t > 0.5, 1.0 - 0.6 * (t - 0.5), 1.0)
    g = np.clip(0.30 + 1.2 * t, 0, 0.9) * np.where(t > 0.5, 1.0 - 1.8 * (t - 0.5), 1.0)
    b = np.clip(0.75 + 0.5 * t, 0, 1) * np.where(t > 0.5, 1.0 - 1.6 * (t - 0.5), 1.0)
    return np.stack([r, g, b, t], axis=1).astype(f32)


IDENTITY_XFM = dict(vx=[1, 0, 0], vy=[0, 1, 0], vz=[0, 0, 1], p=[0, 0, 0])


def default_settings():
    """exa/viewer.cpp:72-80,113-115,124,466 defaults (AO off as the cmdline default)."""
    return dict(dt=0.5, xfOpacityScale=1.0, gradientShadingDVR=1, gradientShadingISO=1,
                spaceSkipping=1, ao_enabled=0, ao_length=1e20)


def fill_frame_state(fs, cam, xf_domains, xfOpacityScale=1.0, frameID=0, iso=None, clip=None,
                     ao_enabled=0, ao_length=1e20, xfm=None, contour=None):
    """fill a ctypes FrameState-like struct (oracle's or the C ABI's — same field names)."""
    for k, dst in (("pos", fs.cam_pos), ("dir00", fs.cam_dir00), ("dirDu", fs.cam_dirDu), ("dirDv", fs.cam_dirDv)):
        for i in range(3):
            dst[i] = float(cam[k][i])
    for i in range(2):
        fs.iso[i].enabled, fs.iso[i].value, fs.iso[i].channel = 0, 0.0, 0
    for i, spec in enumerate(iso or []):
        fs.iso[i].enabled, fs.iso[i].value, fs.iso[i].channel = 1, float(spec[0]), int(spec[1])
    for i in range(3):
        fs.contour[i].enabled = 0
        fs.contour[i].normal[0], fs.contour[i].normal[1], fs.contour[i].normal[2] = 1.0, 0.0, 0.0
        fs.contour[i].channel, fs.contour[i].offset = 0, 0.5
    for i, (normal, offset, channel) in enumerate(contour or []):
        n = _normalize(normal)            # OptixRenderer::updateContourPlanes normalises (OptixRenderer.cpp:511)
        fs.contour[i].enabled = 1
        fs.contour[i].normal[0], fs.contour[i].normal[1], fs.contour[i].normal[2] = float(n[0]), float(n[1]), float(n[2])
        fs.contour[i].channel, fs.contour[i].offset = int(channel), float(offset)
    fs.clipBox.enabled = 0
    if clip is not None:
        fs.clipBox.enabled = 1
        for i in range(3):
            fs.clipBox.lo[i], fs.clipBox.hi[i] = float(clip[0][i]), float(clip[1][i])
    fs.ao.enabled, fs.ao.length = int(ao_enabled), float(ao_length)
    fs.clockScale = 0.0
    x = xfm or IDENTITY_XFM
    for i in range(3):
        fs.xfm_vx[i], fs.xfm_vy[i], fs.xfm_vz[i], fs.xfm_p[i] = (float(x["vx"][i]), float(x["vy"][i]),
                                                               float(x["vz"][i]), float(x["p"][i]))
    fs.frameID = int(frameID)
    for c in range(10):
        fs.xfDomain[c][0], fs.xfDomain[c][1] = 0.0, 1.0
    for c, d in enumerate(xf_domains):
        fs.xfDomain[c][0], fs.xfDomain[c][1] = float(d[0]), float(d[1])
    fs.xfOpacityScale = float(xfOpacityScale)
    return fs


def unpack_rgba8(rgba):
    a = np.asarray(rgba, dtype=np.uint32)
    return np.stack([(a >> 0) & 255, (a >> 8) & 255, (a >> 16) & 255, (a >> 24) & 255], axis=-1).astype(np.uint8)


def write_png(path, rgba, flip=True):
    """tiny dependency-free PNG writer (row 0 of the framebuffer is the bottom row)."""
    import struct
    import zlib
    img = unpack_rgba8(rgba)
    if flip:
        img = img[::-1]
    H, W, _ = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(H))

    def chunk(t, d):
        c = struct.pack(">I", len(d)) + t + d
        return c + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 6, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


# ---- image-space sharding layout (host mirror of exa_hip_set_shard / exa_hip_untile) ----
TILE = 16


def shard_stride(W, H, world):
    tiles = ((W + TILE - 1) // TILE) * ((H + TILE - 1) // TILE)
    return ((tiles + world - 1) // world) * TILE * TILE


def shard_from_image(img, rank, world):
    """tile-major buffer of the 16x16 tiles t with t % world == rank (row-major tile order)."""
    H, W = img.shape
    tx, ty = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    out = np.zeros(shard_stride(W, H, world), dtype=img.dtype)
    for t in range(rank, tx * ty, world):
        x0, y0 = (t % tx) * TILE, (t // tx) * TILE
        blk = np.zeros((TILE, TILE), dtype=img.dtype)
        sub = img[y0:y0 + TILE, x0:x0 + TILE]
        blk[:sub.shape[0], :sub.shape[1]] = sub
        s = (t // world) * TILE * TILE
        out[s:s + TILE * TILE] = blk.ravel()
    return out


def untile(gathered, W, H, world):
    """inverse of shard_from_image over the concatenated shards of all ranks."""
    stride = shard_stride(W, H, world)
    tx, ty = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    img = np.zeros((H, W), dtype=gathered.dtype)
    for t in range(tx * ty):
        x0, y0 = (t % tx) * TILE, (t // tx) * TILE
        s = (t % world) * stride + (t // world) * TILE * TILE
        blk = gathered[s:s + TILE * TILE].reshape(TILE, TILE)
        img[y0:y0 + TILE, x0:x0 + TILE] = blk[:min(TILE, H - y0), :min(TILE, W - x0)]
    return img
