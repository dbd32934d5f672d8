"""Sweep of seeded random cases for the properties that must hold whatever the launch looks like (no oracle involved):
    python tests/gpu_fuzz_sched.py FIRST LAST [--keep-going]
A case of tests/fuzz_cases.py at a frame of 8..60 tiles, rendered (2 accumulated frames) with the defaults, then again
with a random launch order, with the cost feedback in its three phases, with every tile on the wide march (2 and 4 lanes
per ray; one primary channel only), with the memory layouts and launch plans of round 3 toggled (brick order, channel
interleaving, 64-bit addresses, the split pre-pass plan), with either walk of the kd path and the AO rays in front of or beside the march (round 5), as the shards of a random number of ranks re-assembled by the device untile kernel
and by its host mirror, and through a multi-device handle: every one must give the same RGBA8 frame and the same
accumulation buffer bit for bit."""
import sys
import time

import numpy as np

from fuzz_cases import random_case, random_rich_case


def _same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


def _frames(R, n=2):
    out = None
    for f in range(n):
        R.updateFrameID(f)
        out = (np.array(R.render(), copy=True), R.readAccum().copy())
    return out


def check(seed):
    import torch
    from owlexabrick_amd import binding, harness
    rng = np.random.default_rng(0x5C4ED000 + seed)
    if rng.uniform() < 0.5:
        case, desc = random_case(seed, grids=bool(rng.uniform() < 0.3))
    else:
        case, desc, _ = random_rich_case(seed)
    case.W, case.H = int(rng.integers(40, 161)), int(rng.integers(24, 97))
    if isinstance(case.camera, dict):
        case.camera = None
    bad = []
    R = case.hip_renderer()
    base = _frames(R)
    # launch order, feedback phases
    R.setOption("tile_order", int(rng.integers(0, 8)))
    if not _same(_frames(R), base):
        bad.append("tile_order")
    R.setOption("tile_feedback", 1)
    for k in range(3):
        if not _same(_frames(R), base):
            bad.append(f"feedback frame {k}")
    wide_ok = case.nprim == 1
    if wide_ok:
        for lanes in (2, 4):
            R.setOption("wide_march", lanes)
            if not _same(_frames(R), base):
                bad.append(f"wide_march {lanes}")
        R.setOption("wide_march", 1)
    # round 3: where the cells lie in memory and how a frame with surfaces is launched (round 4: ao_defer 2 = sorted AO rays)
    for key, val in (("brick_order", 1), ("interleave", 0), ("addr64", 1), ("prepass_split", 0), ("brick_order", 0), ("interleave", 1),
                     ("prepass_split", 1), ("addr64", 0), ("ao_defer", 0), ("ao_defer", 2), ("ao_defer", 1),
                     # round 5: which walk the march takes, and whether the AO rays run beside it
                     ("walk", 1), ("walk", 2), ("ao_overlap", 0), ("walk", 0), ("ao_overlap", 1),
                     # ... and region ids instead of packed region records in the walk's leaf references, with either walk
                     ("pack_records", 0), ("walk", 1), ("walk", 2), ("pack_records", 1), ("ao_overlap", 0), ("walk", 0)):
        R.setOption(key, val)
        for k in range(2 if key == "prepass_split" else 1):      # the plan changes after its measuring frame
            if not _same(_frames(R), base):
                bad.append(f"{key}={val} (pass {k})")
    R.close()
    # shards of a random world, re-assembled
    world = int(rng.choice([2, 3, 5, 8, 13]))
    W, H = case.W, case.H
    stride = harness.shard_stride(W, H, world)
    gathered = torch.zeros(stride * world, dtype=torch.int32, device="cuda")
    root = None
    for rank in range(world):
        Rk = case.hip_renderer()
        Rk.setShard(rank, world)
        Rk.setOption("tile_feedback", int(rng.integers(0, 2)))
        shard = torch.zeros(stride, dtype=torch.int32, device="cuda")
        for f in range(2):
            Rk.updateFrameID(f)
            Rk.render(device_ptr=shard.data_ptr())
        gathered[rank * stride:(rank + 1) * stride] = shard
        if rank == 0:
            root = Rk
        else:
            Rk.close()
    out = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    root.untile(gathered.data_ptr(), stride, world, out.data_ptr())
    torch.cuda.synchronize()
    root.close()
    if not np.array_equal(out.cpu().numpy().view(np.uint32).reshape(H, W), base[0]):
        bad.append(f"shards of {world} + device untile")
    if not np.array_equal(harness.untile(gathered.cpu().numpy().view(np.uint32), W, H, world), base[0]):
        bad.append(f"shards of {world} + host untile")
    # one handle over several devices (all of them device 0 here)
    ndev = int(rng.choice([2, 3, 4]))
    orig = binding.Renderer

    class Multi(orig):
        def __init__(self, prep, device=0, multiFieldDvr=True):
            super().__init__(prep, multiFieldDvr=multiFieldDvr, devices=[0] * ndev)
    binding.Renderer = Multi
    try:
        Rm = case.hip_renderer()
    finally:
        binding.Renderer = orig
    if not _same(_frames(Rm), base):
        bad.append(f"multi-device handle over {ndev}")
    Rm.close()
    desc.update(W=W, H=H, world=world, ndev=ndev)
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
