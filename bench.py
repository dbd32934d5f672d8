#!/usr/bin/env python3
"""bench.py — frames/s of the ExaBrick DVR hot path on MI355X.

One step = one frame: region kd-tree walk + adaptive ray march + compositing of
every pixel of a 2048x2048 frame of the exajet-like scene (BASELINE.json
configs[3] as a seeded procedural stand-in), scene resident in HBM.
With N GPUs the frame is split into interleaved 16x16 tiles, one process per GPU,
and the RGBA8 tiles are gathered to rank 0 over RCCL (strong scaling: the frame
is fixed).  Prints ONE JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scale S] [--size PX]

`python bench.py --gpus N` starts its own N ranks (fresh child processes, started before this
process imports torch or touches a GPU); under `python -m torch.distributed.run --nproc-per-node N`
(RANK / WORLD_SIZE in the environment) it is one of the ranks.
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
    scalar actually read, per segment the 44 B region record + node_bytes per node of the structure
    that was walked, per pixel 4 B RGBA8 + 16 B accum write (+16 B accum read after frame 0)."""
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


# HIP maps streams onto a few hardware queues (default 4).  One frame's launches go to up to three side streams of the
# module plus the caller's stream; bench adds a communication stream and RCCL its own.  Two of the march streams on one
# hardware queue serialise the frame's kernels (measured: a shard of 8 went from 4.6 to 7.0 ms), so ask for more queues.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def spawn_ranks(n):
    """`bench.py --gpus N` without a launcher: start N fresh rank processes and wait for them.  Called before this
    process has imported torch or made any GPU call; the children are new interpreters (never an exec of a process
    that has touched the GPU).  Rank 0 prints the JSON line on the inherited stdout.  A failed rank ends the others
    and the exit code is non-zero; nothing is retried."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), EXA_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL between processes)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                print(f"[bench] rank process {procs.index(p)} (pid {p.pid}) exited with {code}; stopping the others",
                      file=sys.stderr, flush=True)
                for q in live:
                    q.terminate()                                # exact PIDs of our own children
    return rc if rc >= 0 else 1


# the kernels of one frame: the shipped march (STATS template argument 0), its wide variants for critical tiles, and the
# surfaces pre-pass (iso-surfaces, AO); the counting variants (STATS 1 / 2) are not part of a timed frame
_B = "(true|false)"
FRAME_KERNELS = {"march": rf"renderFrameKdKernel<{_B}, {_B}, \d, {_B}, 0, {_B}(, \d)?(, {_B})?>|renderFrameKernel<{_B}, {_B}, 0>",   # <GRAD, FAST, MULTI, SURF, STATS, SMALL, NCH, ROPE> / <GRAD, ISO, STATS>
                 "march_wide": r"renderFrameKdWideKernel<",
                 "surfaces_prepass": rf"surfacePrepassKdKernel<0, {_B}, {_B}>",                                          # <STATS, ISO_ONLY, AO_DEFER>
                 "ao_rays": r"aoRaysKdKernel<"}
PMC_MARKER = "profileMarkerKernel"           # bench.py brackets its timed frames with two of these (option profile_marker)
PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE", "SQ_INSTS_VALU"))     # HBM reads alone (MI355X_MICROARCH.md: separate passes)
VALU_PEAK_GINSTR = 256 * 4 * 2.4 / 2.0       # G wave64-VALU instructions / s: 1024 SIMDs, one every 2 cycles, 2.4 GHz
F32_VECTOR_PEAK_TFLOPS = 157.3               # MI355X_MICROARCH.md: fp32 vector peak (FMA counted as 2)


class PmcBracketError(RuntimeError):
    pass


def pmc_frame_totals(csv_files, classes=FRAME_KERNELS, frames=None, need_bracket=False):
    """rocprofv3 counter_collection CSVs -> ({kernel class: {counter: mean per FRAME}}, frames).  A dispatch's value is the
    sum of its rows (one row per counter instance: XCD, channel, ...).  With two marker dispatches in the list (PMC_MARKER)
    only the dispatches between them count — the timed frames, without warm-up, cost-measurement and counting frames —
    and `frames` is how many frames the run put there (the caller knows: steps x spp; a frame may launch the march more
    than once — the split pre-pass plan, wide tiles — so dispatches cannot be counted instead).  Without markers every
    dispatch counts and a frame is one dispatch of the "march" class (only true for frames that launch the march once);
    need_bracket turns a missing bracket, or a missing frame count, into a PmcBracketError instead of that fallback."""
    import csv
    import re
    rows = []
    for f in csv_files:
        with open(f, newline="") as fh:
            rows += [(f, int(r["Dispatch_Id"]), r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])) for r in csv.DictReader(fh)]
    marks = {}
    for f, did, name, _, _ in rows:
        if PMC_MARKER in name:
            marks.setdefault(f, set()).add(did)
    bracket = {f: (min(m), max(m)) for f, m in marks.items() if len(m) >= 2}
    if need_bracket and not (bracket and frames):
        raise PmcBracketError(f"the timed frames are not bracketed: {sum(len(m) for m in marks.values())} marker dispatch(es) "
                              f"in {len(csv_files)} file(s), frames={frames}")
    tot, disp = {}, {}
    for f, did, name, counter, value in rows:
        if bracket and not (f in bracket and bracket[f][0] < did < bracket[f][1]):
            continue
        for cls, rx in classes.items():
            if re.search(rx, name):
                d = tot.setdefault(cls, {})
                d[counter] = d.get(counter, 0.0) + value
                disp.setdefault((cls, counter), set()).add((f, did))
                break
    if not (bracket and frames):
        frames = max((len(v) for (cls, _), v in disp.items() if cls == "march"), default=0)
    if not frames or "march" not in tot:
        return {}, 0
    return {cls: {c: v / frames for c, v in d.items()} for cls, d in tot.items()}, frames


def live_pmc(child_args, seconds=300.0, extra_env=None, frames=None):
    """HBM bytes and VALU wave-instructions per frame, by kernel class, measured in THIS run: for each counter group a child
    `rocprofv3 --pmc ... -- python3 bench.py <same workload> --steps 2` (counters only, no trace domains; the parent is
    idle meanwhile).  Returns ({class: {counter: mean per frame}}, frames) or (None, reason)."""
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out_dir = tempfile.mkdtemp(prefix="exa_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", EXA_BENCH_CPU_THREADS="2", EXA_BENCH_NO_LATENCY="1", EXA_BENCH_MARKERS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
              "EXA_BENCH_FORCE_DIST", "EXA_BENCH_SHARD", "EXA_BENCH_SPAWNED", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env.update(extra_env or {})
    merged = {}
    try:
        for i, counters in enumerate(PMC_PASSES):
            cmd = [exe, "--pmc", *counters, "--output-format", "csv", "-d", os.path.join(out_dir, f"pass{i}"), "--",
                   sys.executable, os.path.abspath(__file__), *child_args]
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                 start_new_session=True)
            # A counter request the hardware cannot serve is refused by the profiler's tool library at the first dispatch
            # ("Could not construct profile cfg ... error code 38"); the library then aborts the child and hangs in its own
            # finalisation (recorded in round 4: gpurun_out/r04_e_ta/pass0.log — blocked in the profiler, not in bench.py),
            # so the child would sit there until the time limit.  Watch its stderr and end it at once instead.
            err_lines, refused = [], []

            def watch():
                for raw in iter(p.stderr.readline, b""):
                    line = raw.decode(errors="replace").rstrip()
                    err_lines.append(line)
                    if "Could not construct profile cfg" in line or "rocprofv3 caught signal" in line:
                        refused.append(line)
                        try:
                            os.killpg(p.pid, signal.SIGKILL)             # the process group this call created
                        except ProcessLookupError:
                            pass
                        return
            watcher = threading.Thread(target=watch, daemon=True)
            watcher.start()
            try:
                p.wait(timeout=seconds)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                p.wait()
                return None, f"pass {i} ({' '.join(counters)}) exceeded {seconds:.0f}s"
            watcher.join(timeout=5.0)
            if refused:
                return None, f"pass {i} ({' '.join(counters)}) refused by the profiler: {refused[0][:200]}"
            if p.returncode != 0:
                tail = err_lines[-1:] or [""]
                return None, f"pass {i} ({' '.join(counters)}) exited with {p.returncode}: {tail[0][:200]}"
            csvs = glob.glob(os.path.join(out_dir, f"pass{i}", "**", "*counter_collection.csv"), recursive=True)
            if os.environ.get("EXA_BENCH_KEEP_PMC_CSV"):             # keep the raw lists (tests/golden fixtures, debugging)
                for n_, f_ in enumerate(csvs):
                    shutil.copy(f_, os.path.join(os.environ["EXA_BENCH_KEEP_PMC_CSV"], f"pass{i}_{n_}_counter_collection.csv"))
            try:
                # the child was told to bracket its timed frames (EXA_BENCH_MARKERS): no bracket = no figure, not a guess
                tot, frames = pmc_frame_totals(csvs, frames=frames, need_bracket=True)
            except PmcBracketError as e:
                return None, f"pass {i} ({' '.join(counters)}): {e}"
            for c in counters:
                if c not in tot.get("march", {}):
                    return None, f"pass {i}: no dispatch of the march kernel carries {c}"
            for cls, d in tot.items():
                merged.setdefault(cls, {}).update({c: d[c] for c in counters if c in d})
        return merged, frames
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def useful_flops(st, grad):
    """SURVEY 8(d), informational: ~60 flop per addBasisFunctions call (+~70 with derivatives), ~40 per integrateVolume"""
    return st["brick_visits"] * (60.0 + (70.0 if grad else 0.0)) + st["samples"] * 40.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="timed frames (SURVEY 8(d): 3 warm-up + 50 timed, as viewer.cpp:309)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=float, default=float(os.environ.get("EXA_BENCH_SCALE", "1.0")),
                    help="root-grid scale of the exajet-like scene (1.0 = ~6.4e8 cells)")
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--config", default="c4_exajet")
    ap.add_argument("--camera", default="default", choices=["default", "closeup"],
                    help="default: the viewer's 3/4 view from outside (viewer.cpp:1289-1294); closeup: inside the refined zone")
    ap.add_argument("--fields", type=int, default=None, help="number of scalar fields = DVR channels (default: the configuration's)")
    ap.add_argument("--bricks-file", default=None,
                    help="[n,7] int32 .npy of brick headers (size.xyz, lower.xyz, level): the configuration's scene on these bricks "
                         "instead of the generator's own 8^3 blocks, e.g. profiles/c4_exajet_exabuilder_bricks7.npy = the bricks this "
                         "repo's exaBuilder makes of the scene's cells (tools/builder_bench.py --save-bricks7)")
    ap.add_argument("--no-grad", action="store_true", help="gradient shading off (reference default is on)")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "off"])
    ap.add_argument("--cpu-seconds", type=float, default=45.0)
    ap.add_argument("--pmc", default="auto", choices=["auto", "on", "off"],
                    help="on: rank 0 of a 1-GPU run measures HBM bytes and VALU instructions of the march kernel live (two "
                         "short child runs of this script under rocprofv3 --pmc); auto: the same when the full report is "
                         "being produced (CPU baseline not switched off) and this process is not itself being profiled; "
                         "off / failure: the committed profiles/hbm_traffic.json of the same workload")
    ap.add_argument("--tile-order", type=int, default=int(os.environ.get("EXA_TILE_ORDER", "4")))
    ap.add_argument("--accel", type=int, default=int(os.environ.get("EXA_ACCEL", "1")),
                    help="1 = region kd-tree walked front to back (default), 0 = LBVH restarted per segment")
    ap.add_argument("--basis-form", type=int, default=int(os.environ.get("EXA_BENCH_BASIS_FORM", "-1")), choices=[-1, 0, 1],
                    help="association of the eight-corner basis sums, on the GPU and in the CPU baseline alike: 0 = the reference's "
                         "source order, 1 = per axis with fused multiply-adds (DESIGN.md 2); -1 = the module's default")
    ap.add_argument("--iso", type=float, default=None, help="enable one implicit iso-surface at this value (channel 0)")
    ap.add_argument("--ao", action="store_true", help="ambient-occlusion rays on surface hits (2 per hit, reference default)")
    ap.add_argument("--spp", type=int, default=1, help="frames accumulated per step (frameID 0..spp-1); a step is one converged frame")
    ap.add_argument("--dump", default=None, help="write the frame as PNG (rank 0)")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=INT", help="exa_hip_set_option (tuning knobs)")
    ap.add_argument("--in-flight", type=int, default=int(os.environ.get("EXA_BENCH_IN_FLIGHT", "0")),
                    help="frames in flight: F > 1 renders consecutive frames with F renderer handles on F streams, so that "
                         "the tail of one frame (its longest rays) overlaps the bulk of the next; every frame is complete "
                         "and gathered, frame k+F waits for frame k.  0 = default: 4 for every number of GPUs (one protocol "
                         "for the whole scaling curve: on one GPU, which a frame fills anyway, it is worth about 1 %%; on a "
                         "shard of 8 a rank's frame does not fill its GPU: 4.08 ms with one frame at a time, 2.54 / 2.34 / 2.23 / "
                         "2.27 with 2 / 3 / 4 / 6 in flight, rehearsed); `latency_ms` "
                         "in the line is one frame at a time")
    ap.add_argument("--spawn-check", action="store_true",
                    help="every rank prints its RANK/WORLD_SIZE/MASTER_* as one JSON line and exits (no GPU; tests)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.spawn_check:
        print(json.dumps({"rank": rank, "world": world, "local_rank": local_rank, "pid": os.getpid(), "ppid": os.getppid(),
                          "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}",
                          "torch_imported": "torch" in sys.modules}), flush=True)
        if os.environ.get("EXA_BENCH_FAIL_RANK") == str(rank):
            sys.exit(7)
        if os.environ.get("EXA_BENCH_FAIL_RANK"):
            time.sleep(30)                      # a healthy rank waiting in a collective: the parent must end it
        return
    if world != args.gpus:
        # the launcher's world size is what runs; say so instead of printing a line that claims --gpus
        log(f"warning: WORLD_SIZE={world} != --gpus {args.gpus}; this is a {world}-rank run")

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
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} HIP device(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # EXA_BENCH_FORCE_DIST=1: a 1-rank job still goes through the process group, the gather and the stream pipeline
    # (lets a one-GPU box run the RCCL calls of the N-rank path)
    use_dist = world > 1 or bool(os.environ.get("EXA_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    n_ranks_seen = dist.get_world_size() if use_dist else 1
    cdev = dev if backend == "nccl" else torch.device("cpu")     # where collectives operate

    W = H = args.size
    host_threads = max(2, effective_cpus() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))))

    # ---------------- scene: generate, prepare, upload ----------------
    t0 = time.time()
    ext_bricks = np.load(args.bricks_file) if args.bricks_file else None
    scene = scenes.config(args.config, scale=args.scale, threads=host_threads, fields=args.fields, bricks7=ext_bricks)
    t_gen = time.time() - t0
    log(f"scene {args.config} scale {args.scale}: {scene.num_cells:.4g} cells, {scene.bricks7.shape[0]} bricks, "
        f"levels {scene.meta['levels_hist']} ({t_gen:.1f}s)")

    # the CPU baseline needs the oracle's own scene (its own region build, serial C);
    # start it now on one core so it overlaps the GPU part
    oracle_box = {}
    want_cpu = (args.cpu_baseline == "auto" and rank == 0 and world == 1 and args.iso is None and args.spp == 1
                and len(scene.fields) == 1 and not os.environ.get("EXA_BENCH_SHARD"))     # (a rehearsed shard is not a frame)

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
    lo, hi = prep.voxel_bounds()
    cam = (harness.closeup_camera if args.camera == "closeup" else harness.default_camera)(lo, hi, W, H)   # exa/viewer.cpp:1289-1294
    xf = harness.default_xf()
    # EXA_BENCH_SHARD="r,w" (with EXA_BENCH_FORCE_DIST=1 on a one-GPU box): this single rank renders what rank r of a
    # w-GPU job renders and sends its shard through the process group — a rehearsal of one rank's per-frame work
    rehearse = os.environ.get("EXA_BENCH_SHARD") if world == 1 else None
    shard_rank, shard_world = (int(x) for x in rehearse.split(",")) if rehearse else (rank, world)

    F = args.in_flight if args.in_flight > 0 else 4          # the same protocol for every N: `value` is a throughput
    basis_form = args.basis_form if args.basis_form >= 0 else binding.DEFAULT_BASIS_FORM

    def make_renderer():
        R = binding.Renderer(prep, device=local_rank)
        R.resizeFrameBuffer((W, H))
        R.setOption("tile_order", args.tile_order)
        R.setOption("accel", args.accel)
        if F > 1:
            # the wide march shortens a lone frame's critical path at the price of extra work; with frames in flight
            # the next frame fills the GPU instead and the extra work only costs (rank 0 of 8: 2.9 vs 4.6 ms per frame)
            R.setOption("wide_march", 0)
        R.setOption("basis_form", basis_form)
        for kv in args.option:
            k, v = kv.split("=")
            R.setOption(k, int(v))
        R.setShard(shard_rank, shard_world)
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
        return R

    t0 = time.time()
    Rs = [make_renderer() for _ in range(F)]       # F > 1: one handle (scene copy, accumulation buffer, streams) per frame in flight
    R = Rs[0]
    t_up = time.time() - t0
    log(f"prep {t_prep:.1f}s ({prep.scene.numRegions} regions, {prep.scene.leafListSize} leaf entries), "
        f"upload {t_up:.1f}s ({F} handle(s))")

    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    stride = ((tiles + shard_world - 1) // shard_world) * 256 if shard_world > 1 else W * H   # a 1-rank "shard" is the row-major frame
    # Two shard buffers: while the shard of frame k travels to rank 0, frame k+1 is already being marched into the
    # other one (the accumulation buffer is not read at frameID 0, and with spp > 1 the gather waits for the last
    # sample anyway).  EXA_BENCH_PIPELINE=0 keeps every frame synchronous, like owlLaunch2D.
    pipelined = use_dist and backend == "nccl" and os.environ.get("EXA_BENCH_PIPELINE", "1") != "0"
    shards = [torch.zeros(stride, dtype=torch.int32, device=dev) for _ in range(max(F, 2 if pipelined else 1))]
    final = torch.zeros(W * H if not rehearse else stride, dtype=torch.int32, device=dev) if (use_dist and rank == 0) else None
    # rank 0 receives every shard straight into its slice of one flat buffer (no concatenation step)
    gathered_flat = torch.zeros(stride * world, dtype=torch.int32, device=dev) if (use_dist and rank == 0) else None
    gathered = list(gathered_flat.chunk(world)) if gathered_flat is not None else None
    comm_stream = torch.cuda.Stream(device=dev) if (pipelined or (F > 1 and use_dist)) else None
    consumed = [torch.cuda.Event() for _ in shards]
    march_streams = [torch.cuda.Stream(device=dev) for _ in range(F)] if F > 1 else None
    rendered = [torch.cuda.Event() for _ in shards]
    frame_no = [0]

    def untile(stream_handle):
        if world > 1:
            R.untile(gathered_flat.data_ptr(), stride, world, final.data_ptr(), stream=stream_handle)
        else:
            final.copy_(gathered_flat, non_blocking=True)             # forced 1-rank job: already row-major

    def step():
        k = frame_no[0] % len(shards)
        frame_no[0] += 1
        shard = shards[k]
        if F > 1:
            # F frames in flight: frame k goes to handle k % F on that handle's stream and the call returns at once;
            # the host only waits for frame k - F (its shard gathered), so the GPU always holds the tail of one frame
            # and the bulk of the next
            consumed[k].synchronize()
            Rk, st_k = Rs[k], march_streams[k]
            for f in range(args.spp):
                if args.spp > 1:
                    Rk.updateFrameID(f)
                Rk.render(device_ptr=shard.data_ptr(), stream=st_k.cuda_stream, async_=True)
            if not use_dist:
                consumed[k].record(st_k)
                return
            rendered[k].record(st_k)
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(rendered[k])
                if backend == "nccl":
                    dist.gather(shard, gathered, dst=0)
                else:
                    host = shard.cpu()
                    hl = [torch.zeros_like(host) for _ in range(world)] if rank == 0 else None
                    dist.gather(host, hl, dst=0)
                    if rank == 0:
                        gathered_flat.copy_(torch.cat(hl))
                if rank == 0:
                    untile(comm_stream.cuda_stream)
                consumed[k].record(comm_stream)
            return
        if not pipelined:
            stream = torch.cuda.current_stream().cuda_stream
            for f in range(args.spp):                                      # viewer.cpp:279-288, one launch per sample
                if args.spp > 1:
                    R.updateFrameID(f)
                R.render(device_ptr=shard.data_ptr(), stream=stream)       # synchronous, like owlLaunch2D
            if not use_dist:
                return
            if backend == "nccl":
                dist.gather(shard, gathered, dst=0)           # each peer -> root over its own xGMI link
            else:
                host = shard.cpu()
                hl = [torch.zeros_like(host) for _ in range(world)] if rank == 0 else None
                dist.gather(host, hl, dst=0)
                if rank == 0:
                    gathered_flat.copy_(torch.cat(hl))
            if rank == 0:
                untile(stream)
            return
        # pipelined: the host waits for this frame's march (synchronous render, like owlLaunch2D), hands the shard to the
        # communication stream and starts the next frame's march at once: the gather + untile of frame k run while
        # frame k+1 is marched.  (Queuing the march itself ahead of time was measured and is slower: a frame with wide
        # tiles forks over three streams, and resolving those cross-queue dependencies on the device instead of on the
        # host cost 0.9 ms per frame on a shard of 8 — 5.5 instead of 4.6 ms.)
        consumed[k].synchronize()                                          # the buffer's previous gather is done (long ago)
        stream = torch.cuda.current_stream().cuda_stream
        for f in range(args.spp):
            if args.spp > 1:
                R.updateFrameID(f)
            R.render(device_ptr=shard.data_ptr(), stream=stream)
        with torch.cuda.stream(comm_stream):
            dist.gather(shard, gathered, dst=0)
            if rank == 0:
                untile(comm_stream.cuda_stream)
            consumed[k].record(comm_stream)

    # work counters of this frame (instrumented kernel variant, frameID 0)
    R.updateFrameID(0)
    _, st = R.renderStats()
    log("stats:", {k: v for k, v in st.items() if k not in ("kernel_ms", "rebuild_ms")})
    # the first frames are synchronous in every mode: the launch-order feedback measures tile costs on a
    # synchronous frame and re-orders the launch
    for Rk in Rs:
        for _ in range(2):
            Rk.render(device_ptr=shards[0].data_ptr(), stream=torch.cuda.current_stream().cuda_stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if oracle_thread is not None:
        # the checker's scene build (serial C, ~12 s) overlaps scene upload, counting frame and warm-up, never the timed
        # region: wait for it here
        t_wait = time.perf_counter()
        oracle_thread.join()
        log(f"oracle scene build {oracle_box.get('build_s', 0.0):.1f}s (waited {time.perf_counter() - t_wait:.1f}s for it before the timed region)")
    if use_dist:
        dist.barrier()
    kernel_ms = []
    markers = bool(os.environ.get("EXA_BENCH_MARKERS"))        # a PMC child: bracket the timed frames in the dispatch list
    if markers:
        R.setOption("profile_marker", 1)
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
        if F == 1:
            kernel_ms.append(R.stats()["kernel_ms"])       # the step's last launch (HIP events on the launch stream)
    torch.cuda.synchronize()
    if markers:
        R.setOption("profile_marker", 2)
        torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    el = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    # after the timed region every handle renders one synchronous frame: a synchronous render reads the loop-guard flag
    # (frames queued with async_ do not), so a guard that tripped inside the timed region fails the run here
    torch.cuda.synchronize()
    for Rk in Rs[1:]:
        Rk.render(device_ptr=shards[0].data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    if F > 1:
        # kernel time per launch (HIP events on the launch stream) on synchronous frames after the timed region: with
        # several frames in flight a launch shares the GPU with its neighbours and its own duration says little
        for _ in range(min(5, max(2, args.steps))):
            R.render(device_ptr=shards[0].data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
            kernel_ms.append(R.stats()["kernel_ms"])

    # ---- latency of a lone frame (F = 1): march, gather, untile, host waits — what the reference's synchronous viewer
    #      loop would see.  With one frame in flight the timed region above already is that; with several, measure it
    #      here with the critical tiles marched wide (wide_march = 1, the lone-frame default) ----
    latency_ms = None
    if F > 1 and not os.environ.get("EXA_BENCH_NO_LATENCY"):      # (the PMC children below only need the timed region's frames)
        R.setOption("wide_march", 1)
        cur = torch.cuda.current_stream().cuda_stream
        for _ in range(3):                          # layout changed: cost feedback measures, re-orders, assigns wide tiles
            R.render(device_ptr=shards[0].data_ptr(), stream=cur)

        def lone_frame():
            for f in range(args.spp):
                if args.spp > 1:
                    R.updateFrameID(f)
                R.render(device_ptr=shards[0].data_ptr(), stream=cur)
            if use_dist:
                if backend == "nccl":
                    dist.gather(shards[0], gathered, dst=0)
                else:
                    host = shards[0].cpu()
                    hl = [torch.zeros_like(host) for _ in range(world)] if rank == 0 else None
                    dist.gather(host, hl, dst=0)
                    if rank == 0:
                        gathered_flat.copy_(torch.cat(hl))
                if rank == 0:
                    untile(cur)
            torch.cuda.synchronize()
        lone_frame()
        if use_dist:
            dist.barrier()
        n_lat = max(2, min(args.steps, 20))
        t_l = time.perf_counter()
        for _ in range(n_lat):
            lone_frame()
        if use_dist:
            dist.barrier()
        el_l = torch.tensor([time.perf_counter() - t_l], dtype=torch.float64, device=cdev)
        if use_dist:
            dist.all_reduce(el_l, op=dist.ReduceOp.MAX)
        latency_ms = 1000.0 * float(el_l.item()) / n_lat
        R.setOption("wide_march", 0)

    # the same frame with the basis sums in the reference's source order (option basis_form 0), lone synchronous frames: what
    # the default association (a numerics change within the stated tolerance) buys is then separable in the line
    kernel_ms_form0 = None
    if basis_form == 1 and not os.environ.get("EXA_BENCH_NO_LATENCY"):
        cur = torch.cuda.current_stream().cuda_stream
        R.setOption("basis_form", 0)
        R.updateFrameID(0)
        ks = []
        for i in range(4):
            R.render(device_ptr=shards[0].data_ptr(), stream=cur)
            if i:
                ks.append(R.stats()["kernel_ms"])
        kernel_ms_form0 = float(np.mean(ks)) * args.spp if args.spp == 1 else None
        R.setOption("basis_form", 1)
        R.render(device_ptr=shards[0].data_ptr(), stream=cur)          # shards[0] holds a frame of the default form again

    # aggregate per-rank work counters and kernel time
    agg = torch.tensor([st["samples"], st["brick_visits"], st["corner_loads"], st["segments"], st["nodes_visited"],
                        st["pixels"]], dtype=torch.float64, device=cdev)
    kmax = torch.tensor([float(np.mean(kernel_ms))], dtype=torch.float64, device=cdev)
    if use_dist:
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
    samples_total = float(agg[0].item())

    if rank == 0:
        fps = args.steps / elapsed
        k_ms = float(np.mean(kernel_ms))                   # this rank's launches (HIP events on the launch stream)
        cfg = scenes.CONFIGS[args.config]
        out = {
            "metric": f"frames/sec at {W}^2 DVR{'+iso' if args.iso is not None else ''}{'+AO' if args.ao else ''}"
                      f"{', %d spp' % args.spp if args.spp > 1 else ''}, {args.config.split('_', 1)[1]}-like, MI355X",
            "value": fps, "unit": "frames/s",
            "n_gpus": world, "n_ranks_seen": n_ranks_seen, "steps": args.steps, "warmup": args.warmup,
            **({"rehearsal_of": f"rank {shard_rank} of a {shard_world}-GPU job, on one GPU; value is this rank's frame rate"} if rehearse else {}),
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "frames_in_flight": F,
            "msamples_per_s": samples_total / 1e6 * fps * args.spp,
            "config": {"workload": f"{args.config} (seed {cfg['seed']:#x} procedural AMR, scale {args.scale}"
                                   f"{', bricks from ' + os.path.basename(args.bricks_file) if args.bricks_file else ''}): "
                                   f"{scene.num_cells} cells / {scene.bricks7.shape[0]} bricks / "
                                   f"{int(prep.scene.numRegions)} regions, {W}x{H} DVR"
                                   f"{' of %d channels' % len(scene.fields) if len(scene.fields) > 1 else ''}, "
                                   f"{'camera inside the refined zone, ' if args.camera == 'closeup' else ''}dt 0.5, alpha ramp, "
                                   f"gradient shading {'off' if args.no_grad else 'on'}, space skipping on, frameID 0",
                       "basis_form": f"{basis_form}: " + ("eight-corner basis sums in the reference's source order"
                                                          if basis_form == 0 else
                                                          "eight-corner basis sums associated per axis with fused multiply-adds; the CPU "
                                                          "baseline / crop check evaluates the same sequence (or_set_basis_form)"),
                       "frames_in_flight": (f"{F}: frame k on renderer handle k % {F} (own scene copy, accumulation buffer and stream), "
                                            f"frame k+{F} waits for frame k; every frame is rendered completely and gathered; "
                                            f"kernel_ms is a lone frame's launch") if F > 1 else "1 (every frame synchronous)",
                       "tiling": f"16x16 tiles interleaved over {world} GPU(s)"
                                 + (f", {'RCCL (nccl)' if backend == 'nccl' else backend} gather to rank 0"
                                    f"{', overlapped with the next frame' if pipelined else ''}" if use_dist else ""),
                       "backend": backend if use_dist else None,
                       "samples_per_frame": samples_total, "kernel_ms_max_over_ranks": float(kmax.item())},
            "latency_ms": latency_ms if latency_ms is not None else 1000.0 * elapsed / args.steps,
        }
        if F > 1:
            out["ms_per_step_note"] = (f"elapsed / steps with {F} frames in flight — a throughput: the tail of one frame (its longest rays, few "
                                       "waves busy) overlaps the bulk of the next, so it can be BELOW roofline.kernel_ms, which is one lone "
                                       "synchronous launch, and below latency_ms (one frame at a time); every frame is rendered completely")
        out["config"]["latency_ms"] = ("one frame at a time (march, gather, untile, host waits), wide march on: the figure "
                                       "comparable with a 1-GPU line" if latency_ms is not None else "= ms_per_step (one frame in flight)")
        # ---- roofline: the frame's launches against the resource that binds them.  No dense contraction on
        #      this path (no MFMA) and the working set is cache-resident (HBM far from its peak), what limits the march is
        #      vector-instruction issue: frac = measured wave64 VALU instructions / kernel time against 1024 SIMDs x one
        #      VALU instruction per 2 cycles.  The HBM view (PMC bytes against 8 TB/s) and the useful-flop fraction stand
        #      beside it.  SURVEY 8(d)'s byte formula counts what the LANES REQUEST (most of it served by L1/L2): it is kept
        #      as requested_bytes and never turned into a fraction of the HBM peak. ----
        B = algorithmic_bytes(st, st["pixels"], 0)         # per launch of this rank, frame 0
        flops = useful_flops(st, not args.no_grad)
        roof = {"bound": "valu_issue", "achieved": None, "peak": VALU_PEAK_GINSTR, "unit": "G wave-instr/s", "frac": None,
                "traffic": None,
                "kernel": ("renderFrameKdKernel" if args.accel else "renderFrameKernel")
                          + (" + surfacePrepassKdKernel" if (args.iso is not None and args.accel) else ""),
                "kernel_ms": k_ms,
                **({"kernel_ms_basis_form0": kernel_ms_form0,
                    "kernel_ms_basis_form0_note": "the same frame with option basis_form 0 (basis sums in the reference's source order, nothing "
                                                  "fused), lone synchronous frames after the timed region: kernel_ms / this = what the default "
                                                  "association contributes"} if kernel_ms_form0 is not None else {}),
                "peak_note": "256 CUs x 4 SIMDs x 1 wave64 VALU instruction / 2 cycles at 2.4 GHz (MI355X_MICROARCH.md)",
                "useful_flops_per_launch": flops,
                "useful_flop_frac": flops / (k_ms * 1e-3) / (F32_VECTOR_PEAK_TFLOPS * 1e12),
                "useful_flop_note": f"SURVEY 8(d) flop counts (60 per brick visit + 70 with derivatives, 40 per sample) / kernel time / "
                                    f"{F32_VECTOR_PEAK_TFLOPS} TFLOP/s f32 vector peak (FMA = 2).  The counts are the REFERENCE's operations per call; "
                                    + ("basis_form 1 reaches the same sums with 49 (fused) instead of 116 operations per visit, so this "
                                       "fraction measures delivered reference work, not issued flops"
                                       if basis_form == 1 else "basis_form 0 executes them unfused, one rounding each: 0.5 is its ceiling"),
                "requested_bytes": B,
                "requested_bytes_note": "SURVEY 8(d) formula: bytes the lanes request per launch (36 B per brick visit, 4 B per cell, 44 B "
                                        "per segment, node bytes, 20 B per pixel); served mostly by L1/L2, NOT an HBM figure",
                "requested_bytes_breakdown": {"brick_records_and_leaf_entries": 36 * st["brick_visits"],
                                              "cell_scalars": 4 * st["corner_loads"],
                                              "region_records": 44 * (st["segments"] + st["iso_segments"]),
                                              "accel_nodes": st.get("node_bytes", 64) * st["nodes_visited"],
                                              "framebuffer": 20 * st["pixels"]}}
        out["roofline"] = roof
        traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        key = f"{args.config}@{args.scale}@{W}"
        traffic = vi = None
        stock = (world == 1 and not rehearse and args.iso is None and args.spp == 1 and not args.ao
                 and args.camera == "default" and args.fields is None and not args.no_grad and not args.option
                 and not args.bricks_file)   # what profiles/hbm_traffic.json was measured on
        profiled = any(k.startswith("ROCPROF") for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
        if args.pmc == "on" or (args.pmc == "auto" and args.cpu_baseline == "auto" and not profiled):
            out["_pmc_child"] = True          # measured below, after the process group is gone
        elif os.path.exists(traffic_file) and stock:
            try:
                tf = json.load(open(traffic_file))
                traffic, vi = tf.get(key), tf.get(key + ":valu_wave_instructions")
                roof["pmc_source"] = tf.get(key + ":note")
            except Exception as e:  # noqa: BLE001
                log(f"could not read {traffic_file}: {e}")
        if args.dump:
            img = (final if use_dist else shards[0]).cpu().numpy().view(np.uint32).reshape(H, W)
            harness.write_png(args.dump, img)

        # ---------------- CPU baseline: the oracle on a bounded crop ----------------
        if want_cpu:
            from oracle import pyoracle as po
            S = oracle_box["scene"]                  # built before the timed region (joined there)
            S.set_xf(0, xf)
            S.set_basis_form(basis_form)            # the checker evaluates the same association as the kernels
            fs = po.FrameState()
            harness.fill_frame_state(fs, cam, [scene.value_range], xfOpacityScale=1.0, frameID=0)
            P = po.Params(0.5, 1, 0, 0 if args.no_grad else 1, 1, 1, 1)
            cores = effective_cpus()
            # SURVEY 8(d) protocol: mean of 5 renders after 1 warm-up.  Calibrate on a 64x64 centre window (after a first
            # call that spins the threads up), then size the crop so that the six renders take ~cpu-seconds together
            c0 = W // 2
            S.render(fs, P, W, H, window=(c0 - 8, c0 - 8, c0 + 8, c0 + 8), nthreads=cores)
            t = time.perf_counter()
            _, _, st_c = S.render(fs, P, W, H, window=(c0 - 32, c0 - 32, c0 + 32, c0 + 32), nthreads=cores)
            t_cal = max(time.perf_counter() - t, 1e-4)
            side = int(min(W, max(64, 64 * (args.cpu_seconds / 6.0 / t_cal) ** 0.5))) // 16 * 16
            x0 = (W - side) // 2
            rgba_c, acc_c, st_c = S.render(fs, P, W, H, window=(x0, x0, x0 + side, x0 + side), nthreads=cores)   # warm-up
            t_runs = []
            for _ in range(5):
                t = time.perf_counter()
                S.render(fs, P, W, H, window=(x0, x0, x0 + side, x0 + side), nthreads=cores)
                t_runs.append(time.perf_counter() - t)
            t_cpu = float(np.mean(t_runs))
            # scale by samples (the crop is denser than the frame average), not by pixels
            frame_s = t_cpu * samples_total / max(1, st_c["samples"])
            out["cpu_baseline"] = {"value": 1.0 / frame_s, "unit": "frames/s", "cores": cores, "kind": "port",
                                   "sample": f"{side}x{side} centre crop of the same frame ({st_c['samples']} samples; mean of 5 renders "
                                             f"after 1 warm-up, {t_cpu:.2f}s each on {cores} threads, min {min(t_runs):.2f} max {max(t_runs):.2f}), "
                                             f"scaled to the frame by sample count; oracle scene build {oracle_box['build_s']:.0f}s not included",
                                   "msamples_per_s": st_c["samples"] / 1e6 / t_cpu}
            # the crop doubles as a full-size parity check of the GPU frame
            img = shards[0].cpu().numpy().view(np.uint32).reshape(H, W)
            d = np.abs(harness.unpack_rgba8(img[x0:x0 + side, x0:x0 + side]).astype(int)
                       - harness.unpack_rgba8(rgba_c[x0:x0 + side, x0:x0 + side]).astype(int))
            out["cpu_baseline"]["protocol_note"] = ("SURVEY 8(d) asks for a 512^2 centre crop scaled by PIXEL count for C4/C5; this line scales a "
                                                    f"{side}^2 centre crop by SAMPLE count (the centre of the frame is denser than its average: "
                                                    "pixel scaling would overstate the CPU's frame time); the pixel-scaled figure is given as "
                                                    "value_pixel_scaled")
            out["cpu_baseline"]["value_pixel_scaled"] = 1.0 / (t_cpu * (W * H) / float(side * side))
            out["cpu_baseline"]["crop_max_abs_diff_rgba8"] = int(d.max())
            out["cpu_baseline"]["crop_pixels_differing"] = int((d.max(axis=-1) > 0).sum())
            # the one-thread figure beside the all-threads one (a smaller crop, ~10 s)
            side1 = int(min(side, max(32, side * (10.0 / max(t_cpu * cores, 1e-3)) ** 0.5))) // 16 * 16
            x1 = (W - side1) // 2
            t = time.perf_counter()
            _, _, st_1 = S.render(fs, P, W, H, window=(x1, x1, x1 + side1, x1 + side1), nthreads=1)
            t_1 = time.perf_counter() - t
            out["cpu_baseline"]["one_thread"] = {
                "value": 1.0 / (t_1 * samples_total / max(1, st_1["samples"])), "unit": "frames/s", "cores": 1,
                "msamples_per_s": st_1["samples"] / 1e6 / t_1,
                "sample": f"{side1}x{side1} centre crop ({st_1['samples']} samples, {t_1:.1f}s on 1 thread), scaled by sample count"}

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    for Rk in Rs:
        Rk.close()

    if rank == 0:
        roof = out["roofline"]
        if out.pop("_pmc_child", False):
            # live counters of this very workload, after this process has left the process group and freed its scene
            # copies: fresh children of rank 0 (program after `--` = the python executable) render what THIS rank rendered
            # (its shard of the frame for a multi-GPU job) under rocprofv3 --pmc, counters only
            child = ["--config", args.config, "--scale", str(args.scale), "--size", str(W), "--steps", "2", "--warmup", "1",
                     "--tile-order", str(args.tile_order), "--accel", str(args.accel), "--cpu-baseline", "off", "--pmc", "off",
                     "--in-flight", str(F), "--spp", str(args.spp), "--camera", args.camera]
            child += ["--no-grad"] if args.no_grad else []
            child += ["--basis-form", str(args.basis_form)]
            child += ["--fields", str(args.fields)] if args.fields is not None else []
            child += ["--bricks-file", os.path.abspath(args.bricks_file)] if args.bricks_file else []
            child += ["--iso", str(args.iso)] if args.iso is not None else []
            child += ["--ao"] if args.ao else []
            for kv in args.option:
                child += ["--option", kv]
            extra = {"EXA_BENCH_SHARD": f"{shard_rank},{shard_world}"} if shard_world > 1 else {}
            t = time.perf_counter()
            # the measured line so far goes to stderr first: if a PMC child is interrupted from outside, the frame rate is not lost
            log("result before the live PMC passes (roofline.achieved / frac / traffic pending): " + json.dumps(out))
            vals, n = live_pmc(child, extra_env=extra, frames=2 * args.spp)      # the child times 2 steps of spp frames each
            shard_note = " (the shard of this rank)" if shard_world > 1 else ""
            if vals:
                fetch = sum(d.get("FETCH_SIZE", 0.0) for d in vals.values())
                write = sum(d.get("WRITE_SIZE", 0.0) for d in vals.values())
                # gfx950 counts a 128-B read request as 64 B in FETCH_SIZE (MI355X_MICROARCH.md): 2 * FETCH_SIZE is the upper bound
                traffic = (2.0 * fetch + write) * 1024.0
                vi = sum(d.get("SQ_INSTS_VALU", 0.0) for d in vals.values())
                roof["pmc_source"] = (f"live: rocprofv3 --pmc passes {[' '.join(c) for c in PMC_PASSES]} run by this bench.py on the same "
                                      f"workload{shard_note} ({n} frames averaged, "
                                      f"{time.perf_counter() - t:.0f}s); traffic = 2*FETCH_SIZE({fetch:.6g} KiB) + WRITE_SIZE({write:.6g} KiB)")
                if len(vals) > 1:
                    roof["per_kernel"] = {cls: {"valu_wave_instructions": d.get("SQ_INSTS_VALU"),
                                                "hbm_bytes": (2.0 * d.get("FETCH_SIZE", 0.0) + d.get("WRITE_SIZE", 0.0)) * 1024.0}
                                          for cls, d in vals.items()}
            else:
                log(f"live PMC pass not available ({n})")
                roof["pmc_live_error"] = str(n)
                if os.path.exists(traffic_file) and stock:
                    tf = json.load(open(traffic_file))
                    traffic, vi = tf.get(key), tf.get(key + ":valu_wave_instructions")
                    roof["pmc_source"] = tf.get(key + ":note")
        k_s = roof["kernel_ms"] * 1e-3
        roof["traffic"] = traffic
        # what the march can touch at all (device copies the frame reads): traffic / resident = how often a byte comes from HBM again
        nf, nleaf = len(scene.fields), int(prep.scene.leafListSize)
        res = {"cell_scalars": scene.num_cells * 4 * nf,
               "interleaved_copy_of_the_primary_channels": scene.num_cells * 4 * nf if 2 <= nf <= 4 else 0,
               "march_headers_along_the_leaf_list": nleaf * 32,
               "kd_nodes_and_march_copy": 2 * int(prep.scene.numKdNodes) * 16,
               "region_info": int(prep.scene.numRegions) * 16}
        if st.get("walk_leaf_visits", 0) > 0:
            # the frame took the rope walk: 64 B per leaf (regions + gaps = empty child slots of the kd-tree) and the 16-B nodes
            # behind the links are what it reads instead of the kd-tree's march copy
            kdn = prep.kd_nodes()
            gaps = int((kdn["left"] == binding.KD_EMPTY).sum() + (kdn["right"] == binding.KD_EMPTY).sum()) if len(kdn) else 0
            res["rope_leaves_and_nodes"] = (int(prep.scene.numRegions) + gaps) * 64 + int(prep.scene.numKdNodes) * 16
        roof["hbm_resident_bytes"] = int(sum(res.values()))
        roof["hbm_resident_breakdown"] = res
        if traffic:
            roof["traffic_over_resident"] = traffic / roof["hbm_resident_bytes"]
        if traffic:
            gbs = traffic / k_s / 1e9
            roof["hbm_measured"] = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
            roof["requested_over_fetched"] = roof["requested_bytes"] / traffic
        if vi:
            roof["achieved"] = vi / k_s / 1e9
            roof["frac"] = roof["achieved"] / VALU_PEAK_GINSTR
            roof["valu_wave_instructions_per_launch"] = vi
        else:
            roof["note"] = "no counters for this configuration: achieved / frac are not reported rather than estimated"
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
