#!/usr/bin/env python3
"""A bench point on builder-made bricks: the exajet-like stand-in expanded to the reference's
`.cells` + `.scalars`, bricked by this repo's exaBuilder (same algorithm and flags as builder/builder.cpp:538-811,
bricks up to 127 cells wide, :854-855), loaded through exa::Config by exaRender — next to the generator's own 8^3
bricks of the same cells, written as `.bricks` and loaded the same way.

  python tools/builder_bench.py [--scale 0.35] [--size 2048] [--frames 12] [--out DIR]
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from owlexabrick_amd import scenes  # noqa: E402

HOST = os.path.join(ROOT, "owlexabrick_amd", "host")


def expand_cells(scene):
    """(cells[n,4] int32 x,y,z,level; scalars[n] float32) in brick order, x fastest inside a brick
    (the exajet cell format, builder/builder.cpp:120-134,813-834)"""
    b = np.asarray(scene.bricks7, dtype=np.int64)
    cells = np.empty((scene.num_cells, 4), dtype=np.int32)
    at = 0
    # bricks of equal size are expanded together
    for (sx, sy, sz) in {tuple(r) for r in b[:, 0:3].tolist()}:
        sel = np.nonzero((b[:, 0] == sx) & (b[:, 1] == sy) & (b[:, 2] == sz))[0]
        kz, ky, kx = np.meshgrid(np.arange(sz), np.arange(sy), np.arange(sx), indexing="ij")
        off = np.stack([kx.ravel(), ky.ravel(), kz.ravel()], axis=1)                      # [n,3]
        cw = (1 << b[sel, 6])[:, None, None]
        xyz = b[sel, None, 3:6] + off[None] * cw
        lv = np.broadcast_to(b[sel, 6][:, None, None], xyz.shape[:2] + (1,))
        blk = np.concatenate([xyz, lv], axis=2).astype(np.int32)                          # [bricks, n, 4]
        # position of each brick's cells in the output = its `begin` in brick order
        begins = np.concatenate([[0], np.cumsum(b[:, 0] * b[:, 1] * b[:, 2])])[sel]
        n = sx * sy * sz
        idx = (begins[:, None] + np.arange(n)[None]).ravel()
        cells[idx] = blk.reshape(-1, 4)
        at += len(sel) * n
    assert at == scene.num_cells
    scal = np.asarray(scene.fields[0], dtype=np.float32)[np.asarray(scene.cellIDs)]      # value of the i-th cell
    return cells, scal


def run(cmd, **kw):
    """run to completion; a heartbeat line on stderr every minute (a long exaBuilder run must not look hung)"""
    import tempfile
    t = time.time()
    with tempfile.TemporaryFile("w+") as fo, tempfile.TemporaryFile("w+") as fe:
        p = subprocess.Popen(cmd, stdout=fo, stderr=fe, text=True, **kw)
        while True:
            try:
                p.wait(timeout=60)
                break
            except subprocess.TimeoutExpired:
                print(f"[builder_bench] {os.path.basename(cmd[0])} running, {time.time() - t:.0f}s", file=sys.stderr, flush=True)
        fo.seek(0); fe.seek(0)
        out, err = fo.read(), fe.read()
    if p.returncode:
        raise SystemExit(f"{' '.join(cmd)} failed:\n{out}\n{err}")
    return out, time.time() - t


def render(cfg, size, frames):
    out, _ = run([os.path.join(HOST, "exaRender"), cfg, "--size", str(size), str(size), "--frames", str(frames), "--no-pg",
                  "--range", "0", "1", "--stats"])
    r = {}
    m = re.search(r"bricks (\d+) cells (\d+)", out); r["bricks"], r["cells"] = int(m.group(1)), int(m.group(2))
    m = re.search(r"regions (\d+) leafEntries (\d+)", out); r["regions"], r["leaf_entries"] = int(m.group(1)), int(m.group(2))
    m = re.search(r"stats segments (\d+) samples (\d+) brick_visits (\d+) corner_loads (\d+) nodes_visited (\d+)", out)
    r.update(dict(zip(["segments", "samples", "brick_visits", "corner_loads", "nodes_visited"], map(int, m.groups()))))
    m = re.search(r"Avg. after \d+ frames: ([0-9.]+) FPS \(([0-9.]+) ms\), kernel ([0-9.]+) ms", out)
    r["fps_incl_readback"], r["ms_incl_readback"], r["kernel_ms"] = float(m.group(1)), float(m.group(2)), float(m.group(3))
    r["msamples_per_s_kernel"] = r["samples"] / 1e6 / (r["kernel_ms"] * 1e-3)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.35)
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--out", default="/tmp/exa_builder_bench")
    ap.add_argument("--builder-args", default="", help="extra exaBuilder flags, e.g. '--max-leaf-width 32'")
    ap.add_argument("--save-bricks7", default=None, help="write the builder's brick headers ([n,7] int32: size.xyz, lower.xyz, level) as .npy "
                                                         "(input of bench.py --bricks-file)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    sc = scenes.config("c4_exajet", scale=args.scale)
    res = {"scene": f"c4_exajet scale {args.scale}", "size": args.size, "frames": args.frames}
    # (1) the generator's 8^3 bricks, through the on-disk formats
    print(f"[builder_bench] scene: {sc.num_cells} cells, {sc.bricks7.shape[0]} bricks", file=sys.stderr, flush=True)
    cfg8 = scenes.write_exa(sc, args.out, name="gen8")
    print("[builder_bench] gen8 written", file=sys.stderr, flush=True)
    res["generator_8cubed_bricks"] = render(cfg8, args.size, args.frames)
    # (2) the same cells through exaBuilder
    print("[builder_bench] gen8 rendered; expanding cells", file=sys.stderr, flush=True)
    cells, scal = expand_cells(sc)
    print("[builder_bench] cells expanded; writing + exaBuilder", file=sys.stderr, flush=True)
    cells.tofile(os.path.join(args.out, "built.cells"))
    scal.tofile(os.path.join(args.out, "built_0.scalars"))
    _, t_build = run([os.path.join(HOST, "exaBuilder"), os.path.join(args.out, "built.cells"), "-o",
                      os.path.join(args.out, "built.bricks")] + args.builder_args.split())
    if args.save_bricks7:
        hdrs = []
        with open(os.path.join(args.out, "built.bricks"), "rb") as f:        # size[3], lower[3], level, cellIDs[] per brick
            while True:
                h = f.read(28)
                if len(h) < 28:
                    break
                h = np.frombuffer(h, dtype=np.int32)
                hdrs.append(h.copy())
                f.seek(4 * int(h[0]) * int(h[1]) * int(h[2]), 1)
        np.save(args.save_bricks7, np.stack(hdrs).astype(np.int32))
        print(f"[builder_bench] {len(hdrs)} brick headers -> {args.save_bricks7}", file=sys.stderr, flush=True)
    with open(os.path.join(args.out, "built.exa"), "w") as f:
        f.write("bricks built.bricks\nscalar field0 built_0.scalars\n")
    r = render(os.path.join(args.out, "built.exa"), args.size, args.frames)
    r["exaBuilder_seconds"] = t_build
    r["exaBuilder_args"] = args.builder_args
    res["exaBuilder_bricks"] = r
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
