"""Tests that arm themselves on a box with at least two GPUs and skip on a one-GPU box (round 3): the real peer-store path
of the multi-device handle and the RCCL gather of `bench.py --gpus 2`.  They sit in a file of their own that is collected
last, so that on a box where they run for the first time the rest of the suite has already been through."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from common import Case, ROOT
from test_gpu_multi import CASES, MultiCase, _amr

pytestmark = pytest.mark.gpu


def _ngpu():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("name", ["dvr_grad", "iso_ao", "tiny"])
def test_multi_device_handle_on_two_real_devices_equals_one_device(name):
    """devices [0, 1]: the real peer-store path (hipDeviceEnablePeerAccess, device 1 writes its tiles into device 0's frame)"""
    if _ngpu() < 2:
        pytest.skip("needs two GPUs")
    import torch
    if not torch.cuda.can_device_access_peer(1, 0):
        pytest.skip("device 1 cannot map device 0's memory on this box")
    kw = CASES[name]
    one = Case(_amr(), **kw).run_hip(frames=3, stats=True)
    mc = MultiCase(_amr(), **kw)
    mc.devices = [0, 1]
    multi = mc.run_hip(frames=3, stats=True)
    assert np.array_equal(one[0], multi[0])
    assert np.array_equal(one[1].view(np.uint32), multi[1].view(np.uint32))
    for k in ("segments", "samples", "brick_visits", "corner_loads", "pixels"):
        assert one[2][k] == multi[2][k], k


def test_bench_two_ranks_over_rccl_gives_the_one_rank_frame():
    """bench.py --gpus 2 with the nccl backend: two ranks seen, the gathered frame byte-identical to the 1-rank frame"""
    if _ngpu() < 2:
        pytest.skip("needs two GPUs")
    import json
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with tempfile.TemporaryDirectory() as d:
        lines = {}
        for n in (1, 2):
            png = os.path.join(d, f"f{n}.png")
            r = subprocess.run([sys.executable, bench, "--gpus", str(n), "--steps", "3", "--warmup", "1", "--scale", "0.12", "--size", "512",
                                "--cpu-baseline", "off", "--pmc", "off", "--dump", png], env=env, capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stderr[-2000:]
            lines[n] = (json.loads(r.stdout.strip().splitlines()[-1]), open(png, "rb").read())
        assert lines[2][0]["n_ranks_seen"] == 2 and lines[2][0]["n_gpus"] == 2 and lines[2][0]["config"]["backend"] == "nccl"
        assert lines[1][1] == lines[2][1]
        for k in ("value", "latency_ms", "ms_per_step", "roofline"):
            assert k in lines[2][0], k
