"""One handle over several GPUs (exa_hip_create_multi; SURVEY.md 8(b) "one handle drives N GPUs internally", 8(e)):
the frame is split into interleaved 16x16 tiles, every device renders its tiles on its own stream and stores them
straight into the destination frame of the first device.  On a one-GPU box the device list repeats device 0; the
pixels must be those of the single-device handle, bit for bit."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from common import Case, ROOT, band_xf
from owlexabrick_amd import binding, harness, scenes

pytestmark = pytest.mark.gpu

EXE = os.path.join(ROOT, "owlexabrick_amd", "host", "exaRender")


def _amr():
    return scenes.amr(seed=3, root=(3, 3, 2), B=4, levels=3)


class MultiCase(Case):
    """Case whose HIP renderer is a multi-device handle"""
    devices = [0, 0, 0]

    def hip_renderer(self, device=0):
        orig = binding.Renderer
        devs = self.devices

        class R(orig):
            def __init__(self, prep, device=0, multiFieldDvr=True):
                super().__init__(prep, multiFieldDvr=multiFieldDvr, devices=devs)
        binding.Renderer = R
        try:
            return super().hip_renderer(device)
        finally:
            binding.Renderer = orig


CASES = {
    "dvr_grad": dict(W=200, H=136, grad=1),                                     # ragged tiles
    "band_noskip": dict(W=96, H=96, xf=band_xf(), space_skipping=0),
    "iso_ao": dict(W=96, H=80, grad=1, iso=[(0.45, 0)], ao=1, ao_length=12.0),
    "contour": dict(W=96, H=96, grad=1, opacity_scale=0.05, contour=[([1, 0.3, 0.2], 0.45, 0)]),
    "tiny": dict(W=17, H=5, grad=1),                                            # fewer tiles than devices
}


@pytest.mark.parametrize("ndev", [2, 3, 8])
@pytest.mark.parametrize("name", sorted(CASES))
def test_multi_device_handle_equals_single_device(name, ndev):
    kw = CASES[name]
    one = Case(_amr(), **kw).run_hip(frames=3, stats=True)
    mc = MultiCase(_amr(), **kw)
    mc.devices = [0] * ndev
    multi = mc.run_hip(frames=3, stats=True)
    assert np.array_equal(one[0], multi[0])                                       # RGBA8 frame
    assert np.array_equal(one[1].view(np.uint32), multi[1].view(np.uint32))       # accumulation buffer, 3 frames
    for k in ("segments", "sample_evals", "samples", "brick_visits", "corner_loads", "iso_segments", "iso_evals", "pixels"):
        assert one[2][k] == multi[2][k], k                                        # the devices' work adds up to the frame's


def test_multi_device_async_into_device_buffers_and_state_changes():
    import torch
    case = MultiCase(_amr(), W=104, H=72, grad=1)
    case.devices = [0, 0, 0, 0]
    R = case.hip_renderer()
    ref = Case(_amr(), W=104, H=72, grad=1).hip_renderer()
    bufs = [torch.zeros(104 * 72, dtype=torch.int32, device="cuda") for _ in range(2)]
    stream = torch.cuda.Stream()
    for f in range(4):                                  # alternate buffers, no host wait between the frames
        R.updateFrameID(f)
        R.render(device_ptr=bufs[f & 1].data_ptr(), stream=stream.cuda_stream, async_=True)
    stream.synchronize()
    torch.cuda.synchronize()
    want = None
    for f in range(4):
        ref.updateFrameID(f)
        img = ref.render()
        if f >= 2:
            got = bufs[f & 1].cpu().numpy().view(np.uint32).reshape(72, 104)
            assert np.array_equal(got, img), f
    assert np.array_equal(R.readAccum().view(np.uint32), ref.readAccum().view(np.uint32))
    # setters fan out: a TF edit, an iso-surface, a resize
    for r in (R, ref):
        r.updateXF(0, band_xf()[:, 3], band_xf()[:, :3], case.xf_domains[0], 1.0)
        r.updateIsoValues([0.45, 0], [0, 0], [1, 0])
        r.resizeFrameBuffer((48, 40))
        cam = harness.default_camera(*r.voxelSpaceBounds, 48, 40)
        r.updateCamera(cam["pos"], cam["dir00"], cam["dirDu"], cam["dirDv"])
        r.updateFrameID(0)
    assert np.array_equal(R.render(), ref.render())
    assert np.array_equal(R.readActivity(0), ref.readActivity(0))
    with pytest.raises(RuntimeError, match="shards the frame internally"):
        R.setShard(0, 2)
    R.close()
    ref.close()


def test_exarender_gpus_flag_and_pipelined_copy_out():
    sc = _amr()
    with tempfile.TemporaryDirectory() as d:
        cfg = scenes.write_exa(sc, d, "amr")
        outs = {}
        for tag, extra in (("one", []), ("multi", ["--devices", "0,0,0"]), ("pipe", ["--devices", "0,0", "--pipeline"]),
                           ("pipe1", ["--pipeline"])):
            out = os.path.join(d, tag + ".ppm")
            r = subprocess.run([EXE, cfg, "--size", "120", "88", "--frames", "5", "--no-pg", "--isovals", "0.4", "0.4", "-o", out]
                               + extra, capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr
            assert "Avg. after 5 frames" in r.stdout
            outs[tag] = open(out, "rb").read()
        assert outs["one"] == outs["multi"] == outs["pipe"] == outs["pipe1"]
