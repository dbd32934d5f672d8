#!/usr/bin/env python3
"""bench.py — frames/s of the ExaBrick DVR hot path on MI355X.

One step = one frame: region-LBVH traversal + adaptive ray march + compositing of
every pixel of a 2048x2048 frame of the exajet-like scene (BASELINE.json
configs[3] as a seeded procedural stand-in), scene resident in HBM.
With N GPUs the frame is split into interleaved 16x16 tiles, one process per GPU,
and the RGBA8 tiles are gathered to rank 0 over RCCL (strong scaling: the frame
is fixed).  Prints ONE JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scale S] [--size PX]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def algorithmic_bytes(st, pixels, frame_id=0):
    """per brick visit 32 B record + 4 B leaf-list entry, 4 B per cell
    scalar actually read, per segment the 44 B region record + 64 B per LBVH node
    fetched, per pixel 4 B RGBA8 + 16 B accum write (+16 B accum read after frame 0)."""
    return (36 * (st["brick_visits"]) + 4 * st["corner_loads"]
            + 44 * (st["segments"] + st["iso_segments"]) + st.get("node_bytes", 64) * st["nodes_visited"]
            + pixels * (4 + 16 + (16 if frame_id > 0 else 0)))


def effective_cpus():
    """host cores this process may really use: affinity mask and cgroup quota, not the machine's thread count
    (a one-GPU box of the pool shows 256 hardware threads but grants a share of 16)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    if os.environ.get("EXA_BENCH_CPU_THREADS"):
        n = int(os.environ["EXA_BENCH_CPU_THREADS"])
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=float, default=float(os.environ.get("EXA_BENCH_SCALE", "1.0")),
                    help="root-grid scale of the exajet-like scene (1.0 = ~6.4e8 cells)")
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--config", default="c4_exajet")
    ap.add_argument("--no-grad", action="store_true", help="gradient shading off (reference default is on)")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "off"])
    ap.add_argument("--cpu-seconds", type=float, default=45.0)
    ap.add_argument("--tile-order", type=int, default=int(os.environ.get("EXA_TILE_ORDER", "4")))
    ap.add_argument("--accel", type=int, default=int(os.environ.get("EXA_ACCEL", "1")),
                    help="1 = region kd-tree walked front to back (default), 0 = LBVH restarted per segment")
    ap.add_argument("--iso", type=float, default=None, help="enable one implicit iso-surface at this value (channel 0)")
    ap.add_argument("--ao", action="store_true", help="ambient-occlusion rays on surface hits (2 per hit, reference default)")
    ap.add_argument("--spp", type=int, default=1, help="frames accumulated per step (frameID 0..spp-1); a step is one converged frame")
    ap.add_argument("--dump", default=None, help="write the frame as PNG (rank 0)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log(f"warning: WORLD_SIZE={world} != --gpus {args.gpus}; using WORLD_SIZE")

    import numpy as np
    import torch
    import torch.distributed as dist
    from owlexabrick_amd import binding, harness, scenes

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the render path has no CPU fallback")
    # rehearsal on a one-GPU box: EXA_BENCH_BACKEND=gloo EXA_BENCH_ONE_DEVICE=1 puts every rank on
    # device 0 and gathers through host memory; the measured configuration is nccl (= RCCL), one GPU per rank
    backend = os.environ.get("EXA_BENCH_BACKEND", "nccl")
    if os.environ.get("EXA_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    cdev = dev if backend == "nccl" else torch.device("cpu")     # where collectives operate

    W = H = args.size
    host_threads = max(2, effective_cpus() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))))

    # ---------------- scene: generate, prepare, upload ----------------
    t0 = time.time()
    scene = scenes.config(args.config, scale=args.scale, threads=host_threads)
    t_gen = time.time() - t0
    log(f"scene {args.config} scale {args.scale}: {scene.num_cells:.4g} cells, {scene.bricks7.shape[0]} bricks, "
        f"levels {scene.meta['levels_hist']} ({t_gen:.1f}s)")

    # the CPU baseline needs the oracle's own scene (its own region build, serial C);
    # start it now on one core so it overlaps the GPU part
    oracle_box = {}
    want_cpu = args.cpu_baseline == "auto" and rank == 0 and world == 1 and args.iso is None and args.spp == 1

    def build_oracle():
        from oracle import pyoracle as po          # bench.py's cpu_baseline leg may use the oracle
        t = time.time()
        oracle_box["scene"] = po.OracleScene(scene.bricks7, scene.cellIDs, scene.fields)
        oracle_box["build_s"] = time.time() - t
    oracle_thread = None
    if want_cpu:
        oracle_thread = threading.Thread(target=build_oracle, daemon=True)
        oracle_thread.start()

    t0 = time.time()
    prep = binding.Prep(scene, num_threads=host_threads)
    t_prep = time.time() - t0
    t0 = time.time()
    R = binding.Renderer(prep, device=local_rank)
    t_up = time.time() - t0
    log(f"prep {t_prep:.1f}s ({prep.scene.numRegions} regions, {prep.scene.leafListSize} leaf entries), "
        f"upload {t_up:.1f}s")

    lo, hi = prep.voxel_bounds()
    cam = harness.default_camera(lo, hi, W, H)                 # exa/viewer.cpp:1289-1294
    xf = harness.default_xf()
    R.resizeFrameBuffer((W, H))
    R.setOption("tile_order", args.tile_order)
    R.setOption("accel", args.accel)
    R.setShard(rank, world)
    R.updateCamera(cam["pos"], cam["dir00"], cam["dirDu"], cam["dirDv"])
    R.updateXF(0, xf[:, 3], xf[:, :3], scene.value_range, 1.0)
    for c in range(1, len(scene.fields)):
        R.updateXF(c, xf[:, 3], xf[:, :3], (0.0, 1.0), 1.0)
    if args.iso is not None:
        R.updateIsoValues([args.iso, 0], [0, 0], [1, 0])
    else:
        R.updateIsoValues([0, 0], [0, 0], [0, 0])
    R.setSpaceSkipping(True)
    R.setGradientShadingDVR(not args.no_grad)
    R.updateDt(0.5)
    R.frameState.ao.enabled = 1 if args.ao else 0
    R.updateFrameID(0)

    n_out = R.outputPixels()
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    stride = ((tiles + world - 1) // world) * 256 if world > 1 else W * H
    shard = torch.zeros(stride, dtype=torch.int32, device=dev)
    final = torch.zeros(W * H, dtype=torch.int32, device=dev) if (world > 1 and rank == 0) else None
    # rank 0 receives every shard straight into its slice of one flat buffer (no concatenation step)
    gathered_flat = torch.zeros(stride * world, dtype=torch.int32, device=dev) if (world > 1 and rank == 0) else None
    gathered = list(gathered_flat.chunk(world)) if gathered_flat is not None else None
    mode = {"collective": "gather"}

    def step():
        stream = torch.cuda.current_stream().cuda_stream
        for f in range(args.spp):                                      # viewer.cpp:279-288, one launch per sample
            if args.spp > 1:
                R.updateFrameID(f)
            R.render(device_ptr=shard.data_ptr(), stream=stream)       # synchronous, like owlLaunch2D
        if world == 1:
            return
        if backend == "nccl":
            if mode["collective"] == "gather":
                try:
                    dist.gather(shard, gathered, dst=0)       # each peer -> root over its own xGMI link
                except RuntimeError as e:                     # defensive: fall back to an all-gather
                    log(f"dist.gather failed ({e}); using all_gather_into_tensor")
                    mode["collective"] = "all_gather"
                    mode["flat"] = gathered_flat if rank == 0 else torch.zeros(stride * world, dtype=torch.int32, device=dev)
            if mode["collective"] == "all_gather":
                dist.all_gather_into_tensor(mode["flat"], shard)
        else:
            host = shard.cpu()
            hl = [torch.zeros_like(host) for _ in range(world)] if rank == 0 else None
            dist.gather(host, hl, dst=0)
            if rank == 0:
                gathered_flat.copy_(torch.cat(hl))
        if rank == 0:
            R.untile(gathered_flat.data_ptr(), stride, world, final.data_ptr(), stream=stream)

    # work counters of this frame (instrumented kernel variant, frameID 0)
    R.updateFrameID(0)
    _, st = R.renderStats()
    log("stats:", {k: v for k, v in st.items() if k not in ("kernel_ms", "rebuild_ms")})

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    kernel_ms = []
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(R.stats()["kernel_ms"])       # the step's last launch
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    el = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    # aggregate per-rank work counters and kernel time
    agg = torch.tensor([st["samples"], st["brick_visits"], st["corner_loads"], st["segments"], st["nodes_visited"],
                        st["pixels"]], dtype=torch.float64, device=cdev)
    kmax = torch.tensor([float(np.mean(kernel_ms))], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
    samples_total = float(agg[0].item())

    if rank == 0:
        fps = args.steps / elapsed
        k_ms = float(np.mean(kernel_ms))                   # this rank's launches (HIP events on the launch stream)
        B = algorithmic_bytes(st, st["pixels"], 0)         # per launch of this rank
        achieved = B / (k_ms * 1e-3) / 1e9
        out = {
            "metric": f"frames/sec at {W}^2 DVR{'+iso' if args.iso is not None else ''}{'+AO' if args.ao else ''}"
                      f"{', %d spp' % args.spp if args.spp > 1 else ''}, {args.config.split('_', 1)[1]}-like, MI355X",
            "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "msamples_per_s": samples_total / 1e6 * fps * args.spp,
            "config": {"workload": f"{args.config} (seed 0xE7A0003 procedural AMR, scale {args.scale}): "
                                   f"{scene.num_cells} cells / {scene.bricks7.shape[0]} bricks / "
                                   f"{int(prep.scene.numRegions)} regions, {W}x{H} DVR, dt 0.5, alpha ramp, "
                                   f"gradient shading {'off' if args.no_grad else 'on'}, space skipping on, frameID 0",
                       "tiling": f"16x16 tiles interleaved over {world} GPU(s)" + (", RCCL gather to rank 0" if world > 1 else ""),
                       "samples_per_frame": samples_total, "kernel_ms_max_over_ranks": float(kmax.item())},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "renderFrameKdKernel" if args.accel else "renderFrameKernel", "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": B,
                         "bytes_breakdown": {"brick_records_and_leaf_entries": 36 * st["brick_visits"],
                                             "cell_scalars": 4 * st["corner_loads"],
                                             "region_records": 44 * st["segments"],
                                             "accel_nodes": st.get("node_bytes", 64) * st["nodes_visited"],
                                             "framebuffer": 20 * st["pixels"]}},
        }
        traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(traffic_file):
            try:
                tf = json.load(open(traffic_file))
                out["roofline"]["traffic"] = tf.get(f"{args.config}@{args.scale}@{W}")
                vi = tf.get(f"{args.config}@{args.scale}@{W}:valu_wave_instructions")
                if vi:
                    # what actually bounds the kernel: vector-instruction issue.  A wave64 VALU
                    # instruction holds a SIMD for 2 cycles (MI355X_MICROARCH.md); 256 CUs x 4 SIMDs at 2.4 GHz
                    floor_ms = vi * 2.0 / (256 * 4) / 2.4e9 * 1e3
                    out["roofline"]["valu_issue"] = {"wave_instructions": vi, "floor_ms": floor_ms, "frac": floor_ms / k_ms,
                                                     "source": "SQ_INSTS_VALU, profiles/ (same command under rocprofv3 --pmc)"}
            except Exception:
                pass
        if args.dump:
            img = (final if world > 1 else shard).cpu().numpy().view(np.uint32).reshape(H, W)
            harness.write_png(args.dump, img)

        # ---------------- CPU baseline: the oracle on a bounded crop ----------------
        if want_cpu:
            from oracle import pyoracle as po
            oracle_thread.join()
            S = oracle_box["scene"]
            S.set_xf(0, xf)
            fs = po.FrameState()
            harness.fill_frame_state(fs, cam, [scene.value_range], xfOpacityScale=1.0, frameID=0)
            P = po.Params(0.5, 1, 0, 0 if args.no_grad else 1, 1, 1, 1)
            cores = effective_cpus()
            # calibrate on a 64x64 centre window (after a warm-up that spins the threads up),
            # then size the crop for ~cpu-seconds of work
            c0 = W // 2
            S.render(fs, P, W, H, window=(c0 - 8, c0 - 8, c0 + 8, c0 + 8), nthreads=cores)
            t = time.perf_counter()
            _, _, st_c = S.render(fs, P, W, H, window=(c0 - 32, c0 - 32, c0 + 32, c0 + 32), nthreads=cores)
            t_cal = max(time.perf_counter() - t, 1e-4)
            side = int(min(W, max(64, 64 * (args.cpu_seconds / t_cal) ** 0.5))) // 16 * 16
            x0 = (W - side) // 2
            t = time.perf_counter()
            rgba_c, acc_c, st_c = S.render(fs, P, W, H, window=(x0, x0, x0 + side, x0 + side), nthreads=cores)
            t_cpu = time.perf_counter() - t
            # scale by samples (the crop is denser than the frame average), not by pixels
            frame_s = t_cpu * samples_total / max(1, st_c["samples"])
            out["cpu_baseline"] = {"value": 1.0 / frame_s, "unit": "frames/s", "cores": cores, "kind": "port",
                                   "sample": f"{side}x{side} centre crop of the same frame ({st_c['samples']} samples, "
                                             f"{t_cpu:.1f}s on {cores} threads), scaled to the frame by sample count; "
                                             f"oracle scene build {oracle_box['build_s']:.0f}s not included",
                                   "msamples_per_s": st_c["samples"] / 1e6 / t_cpu}
            # the crop doubles as a full-size parity check of the GPU frame
            img = shard.cpu().numpy().view(np.uint32).reshape(H, W)
            d = np.abs(harness.unpack_rgba8(img[x0:x0 + side, x0:x0 + side]).astype(int)
                       - harness.unpack_rgba8(rgba_c[x0:x0 + side, x0:x0 + side]).astype(int))
            out["cpu_baseline"]["crop_max_abs_diff_rgba8"] = int(d.max())
            out["cpu_baseline"]["crop_pixels_differing"] = int((d.max(axis=-1) > 0).sum())
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    R.close()


if __name__ == "__main__":
    main()
