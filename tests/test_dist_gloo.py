"""The N>1 path on CPU: two gloo ranks shard a frame into interleaved 16x16 tiles,
gather the RGBA8 shards to rank 0 and untile — the same protocol bench.py runs over
RCCL.  The per-rank pixels come from the oracle (tests may use it; there is no CPU
renderer in the product)."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from common import Case, ROOT
from owlexabrick_amd import harness, scenes


def _worker(rank, world, initfile, outfile, W, H):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    case = Case(scenes.example("ex3"), W=W, H=H, grad=1)
    rgba, _, _ = case.run_oracle(nthreads=1)
    # every rank owns only its tiles: blank out the others to prove nothing leaks
    mine = harness.shard_from_image(rgba.astype(np.int64), rank, world)
    shard = torch.from_numpy(mine)
    gathered = [torch.zeros_like(shard) for _ in range(world)] if rank == 0 else None
    dist.gather(shard, gathered, dst=0)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)          # the max-over-ranks timing reduction of bench.py
    assert t.item() == world
    if rank == 0:
        img = harness.untile(torch.cat(gathered).numpy(), W, H, world)
        np.save(outfile, np.stack([img, rgba.astype(np.int64)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("size", [(64, 48), (72, 40)])     # second one has ragged edge tiles
def test_two_rank_tile_gather_reassembles_frame(size):
    W, H = size
    with tempfile.TemporaryDirectory() as d:
        init, out = os.path.join(d, "init"), os.path.join(d, "out.npy")
        mp.spawn(_worker, args=(2, init, out, W, H), nprocs=2, join=True)
        img, ref = np.load(out)
        assert np.array_equal(img, ref)


def test_shard_layout_roundtrip_any_world():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 2 ** 32, size=(50, 70), dtype=np.uint32)
    for world in (1, 2, 3, 4, 8):
        g = np.concatenate([harness.shard_from_image(img, r, world) for r in range(world)])
        assert np.array_equal(harness.untile(g, 70, 50, world), img)
