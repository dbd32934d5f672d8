"""Regenerates tests/golden/ (run in the build container, where /root/reference exists).

1. ref_artificial/exN.{cells,scalars}: OUTPUTS of the reference's own
   tools/artificial generator (built by oracle/ref.mk into oracle/_ref/) on its own
   ex0..ex4.grids inputs — data, not source.
2. oracle_*.npz: frames rendered by the CPU oracle on those inputs (RGBA8, float accum,
   work counters, region table) — ORACLE-derived, not reference-derived.  The reference has no goldens of its own (no tests at
   all), so these pin the oracle against regressions and pin the HIP path on the GPU box.
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from common import Case, band_xf  # noqa: E402
from owlexabrick_amd import scenes  # noqa: E402

GOLDEN_CASES = {
    # name: (scene, kwargs)
    "ex0_dvr": ("ex0", dict(W=48, H=32)),
    "ex1_grad": ("ex1", dict(W=48, H=32, grad=1)),
    "ex2_dvr": ("ex2", dict(W=48, H=32)),
    "ex2_grad_iso": ("ex2", dict(W=48, H=32, grad=1, iso=[(0.2, 0)])),
    "ex3_dvr": ("ex3", dict(W=64, H=48)),
    "ex3_grad": ("ex3", dict(W=64, H=48, grad=1)),
    "ex3_iso": ("ex3", dict(W=64, H=48, grad=1, iso=[(0.4, 0)])),
    "ex4_dvr_band": ("ex4", dict(W=64, H=48, xf="band")),
    "ex4_grad_iso2": ("ex4", dict(W=64, H=48, grad=1, iso=[(0.3, 0), (0.7, 0)])),
    "ex4_accum3": ("ex4", dict(W=64, H=48, grad=1, frames=3)),
    "ex3_contour_iso": ("ex3", dict(W=64, H=48, grad=1, iso=[(0.6, 0)], contour=[([1, 1, 0.2], 0.55, 0)], opacity_scale=0.2)),
    # The full-precision TF filter weight (option tf_filter = 0).  These four files are the goldens as they were BEFORE the
    # 1.8 fixed-point weight became the default (commit c1b5d4e~1, taken from the history, not regenerated): the oracle and
    # the kernels in that mode still give them bit for bit, i.e. the change of default did not move the old path.
    "ex3_grad_tf0": ("ex3", dict(W=64, H=48, grad=1, tf_filter=0)),
    "ex4_dvr_band_tf0": ("ex4", dict(W=64, H=48, xf="band", tf_filter=0)),
    "ex2_grad_iso_tf0": ("ex2", dict(W=48, H=32, grad=1, iso=[(0.2, 0)], tf_filter=0)),
    "ex4_accum3_tf0": ("ex4", dict(W=64, H=48, grad=1, frames=3, tf_filter=0)),
    # BASELINE.json configs[0]: 64^3 single-level brick with the ex2 corner pattern, 512x512, viewer
    # default camera / TF / dt, gradient shading off and on; the fixture holds the centre 192x192 crop of the frame
    "c1_64_512": ("c1_64", dict(W=512, H=512, window=(160, 160, 352, 352))),
    "c1_64_512_grad": ("c1_64", dict(W=512, H=512, grad=1, window=(160, 160, 352, 352))),
    # Everything above is rendered with the basis sums in the reference's source order (basis_form = 0, pinned in
    # make_case).  The per-axis association the kernels ship by default (basis_form = 1) has its own fixtures:
    "ex3_grad_pa": ("ex3", dict(W=64, H=48, grad=1, basis_form=1)),
    "ex3_iso_pa": ("ex3", dict(W=64, H=48, grad=1, iso=[(0.4, 0)], basis_form=1)),
    "ex4_grad_iso2_pa": ("ex4", dict(W=64, H=48, grad=1, iso=[(0.3, 0), (0.7, 0)], basis_form=1)),
    "ex4_accum3_pa": ("ex4", dict(W=64, H=48, grad=1, frames=3, basis_form=1)),
    "c1_64_512_grad_pa": ("c1_64", dict(W=512, H=512, grad=1, window=(160, 160, 352, 352), basis_form=1)),
}


def make_case(name):
    scn, kw = GOLDEN_CASES[name]
    kw = dict(kw)
    frames = kw.pop("frames", 1)
    kw.pop("window", None)
    kw.setdefault("basis_form", 0)
    if kw.get("xf") == "band":
        kw["xf"] = band_xf()
    return Case(scenes.example(scn), **kw), frames


def golden_window(name):
    """(x0, y0, x1, y1) of the crop the fixture holds, or None for the whole frame"""
    return GOLDEN_CASES[name][1].get("window")


def crop(a, window):
    if window is None:
        return a
    x0, y0, x1, y1 = window
    return a[y0:y1, x0:x1]


def main():
    ref = os.path.join(ROOT, "oracle", "_ref", "exaArtificial")
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-f", "ref.mk", "-s"])
        out = os.path.join(HERE, "ref_artificial")
        os.makedirs(out, exist_ok=True)
        for e in ("ex0", "ex1", "ex2", "ex3", "ex4"):
            subprocess.check_call([ref, f"/root/reference/tools/artificial/{e}.grids", "-o", os.path.join(out, e)],
                                  stdout=subprocess.DEVNULL)
    for name in GOLDEN_CASES:
        case, frames = make_case(name)
        win = golden_window(name)
        rgba, acc, st = case.run_oracle(nthreads=1, frames=frames, window=win)
        rgba, acc = crop(rgba, win), crop(acc, win)
        S = case.oracle_scene()
        np.savez_compressed(os.path.join(HERE, f"oracle_{name}.npz"), rgba=rgba, accum=acc,
                            stats=np.array([st[k] for k in sorted(st)], dtype=np.int64),
                            stat_keys=np.array(sorted(st)), regions=S.regions(), leaflist=S.leaflist())
        print(name, st)


if __name__ == "__main__":
    main()
