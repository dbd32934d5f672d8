"""Registers and scratch of the march kernels, read from the gfx950 code objects the build left in csrc/ (no GPU needed).

The march's speed hangs on its occupancy: a variant compiled for seven waves per SIMD must fit 72 vector registers and one
for six waves 80 — and must do so WITHOUT scratch: the same kernel with the pixel's colour spilled inside the march loop takes
22.1 instead of 17.1 ms on the bench scene (profiles/r05_experiments.txt 15), and nothing else in the suite would notice.
Checked here for the variants the default frame of every BASELINE configuration launches (form 1, not instrumented)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "owlexabrick_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernels(obj):
    """{mangled name: (vgprs, scratch bytes, spilled vgprs)} of the gfx950 code object embedded in a host object file"""
    path = os.path.join(CSRC, obj)
    if not (os.path.exists(path) and os.path.exists(os.path.join(LLVM, "llvm-objdump"))):
        pytest.skip(f"{obj} or the llvm tools are not here")
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(path, os.path.join(d, "k.o"))
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "k.o"], cwd=d, check=True, capture_output=True)
        co = [f for f in os.listdir(d) if "gfx950" in f]
        assert co, "no gfx950 code object in " + obj
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co[0]], cwd=d, check=True, capture_output=True,
                               text=True).stdout
    out = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n"
                         r"(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", notes):
        out[m.group(1)] = (int(m.group(3)), int(m.group(2)), int(m.group(4)))
    return out


def _march(ns, grad, fast, multi, surf, stats, small, nch, rope):
    b = lambda v: "Lb1E" if v else "Lb0E"                                    # noqa: E731
    return (f"_ZN3exa{len(ns)}{ns}19renderFrameKdKernelI{b(grad)}{b(fast)}Li{multi}E{b(surf)}Li{stats}E{b(small)}Li{nch}E{b(rope)}"
            "EEvNS_10RenderArgsE")


@pytest.mark.parametrize("surf", [False, True], ids=["dvr", "surfaces"])
@pytest.mark.parametrize("grad", [False, True], ids=["plain", "gradient"])
def test_one_channel_rope_march_fits_seven_waves_without_scratch(grad, surf):
    k = _kernels("exa_kernels_f1r.o")
    # 32-bit address form (every BASELINE configuration): seven waves per SIMD = 72 VGPRs
    vgpr, scratch, spill = k[_march("form1", grad, True, 0, surf, 0, True, 0, True)]
    assert vgpr <= 72 and scratch == 0 and spill == 0, (vgpr, scratch, spill)
    # fields beyond 4 GiB: six waves = 80
    vgpr, scratch, spill = k[_march("form1", grad, True, 0, surf, 0, False, 0, True)]
    assert vgpr <= 80 and scratch == 0 and spill == 0, (vgpr, scratch, spill)


@pytest.mark.parametrize("surf", [False, True], ids=["dvr", "surfaces"])
def test_stack_walk_and_multi_channel_march_have_no_scratch(surf):
    k = _kernels("exa_kernels_f1.o")
    for small in (True, False):
        vgpr, scratch, spill = k[_march("form1", True, True, 0, surf, 0, small, 0, False)]       # six waves
        assert vgpr <= 80 and scratch == 0 and spill == 0, (small, vgpr, scratch, spill)
    kr = _kernels("exa_kernels_f1r.o")
    for nch, budget in ((2, 96), (3, 128), (4, 128)):                                            # five / four / four waves
        vgpr, scratch, spill = kr[_march("form1", True, True, 2, surf, 0, True, nch, True)]
        assert vgpr <= budget and scratch == 0 and spill == 0, (nch, vgpr, scratch, spill)
