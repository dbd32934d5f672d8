#!/usr/bin/env python3
"""CPU-baseline protocol of SURVEY.md 8(d) for the BASELINE.json configurations: the oracle (a scalar C restatement of
programs/exabrick.cu, rows of the image over threads — the reference has no CPU render path of its own) timed on this
host for C1..C3 at full resolution with all threads, and with ONE thread (on a centre crop where a full frame would take
minutes; stated), next to the HIP path on the same frame when a GPU is present.  Prints one JSON object.

  python tools/cpu_baselines.py [--threads N] [--one-thread-seconds 12]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import Case  # noqa: E402  (tools/ may use the oracle: this is the cpu_baseline leg)
from owlexabrick_amd import scenes  # noqa: E402

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def time_oracle(case, S, nthreads, window=None, warm=True):
    fs, P = case.oracle_state(S)
    if warm:
        c = case.W // 2
        S.render(fs, P, case.W, case.H, window=(c - 8, c - 8, c + 8, c + 8), nthreads=nthreads)
    t = time.perf_counter()
    _, _, st = S.render(fs, P, case.W, case.H, window=window, nthreads=nthreads)
    return time.perf_counter() - t, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=bench.effective_cpus())
    ap.add_argument("--one-thread-seconds", type=float, default=12.0)
    ap.add_argument("--configs", default="c1,c2,c3")
    args = ap.parse_args()
    out = {"host_threads": args.threads}
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        have_gpu = False
    cfgs = {
        "c1": ("C1 64^3 single brick, 512x512 DVR", lambda: Case(scenes.example("c1_64"), W=512, H=512, grad=1)),
        "c2": ("C2 LANL-like, 1024x1024 DVR", lambda: Case(scenes.config("c2_lanl"), W=1024, H=1024, grad=1, xf_domains=[(0.0, 1.0)])),
        "c3": ("C3 landing-gear-like, 2048x2048 DVR (2 channels) + iso-surface",
               lambda: Case(scenes.config("c3_gear"), W=2048, H=2048, grad=1, iso=[(0.5, 0)], xf_domains=[(0.0, 1.0)] * 2)),
    }
    for key in args.configs.split(","):
        name, make = cfgs[key]
        case = make()
        S = case.oracle_scene()
        r = {"cells": case.scene.num_cells}
        t_all, st = time_oracle(case, S, args.threads)
        r["all_threads"] = {"threads": args.threads, "ms_per_frame": 1e3 * t_all, "frames_per_s": 1.0 / t_all,
                            "msamples_per_s": st["samples"] / 1e6 / t_all, "samples": st["samples"], "window": "full frame"}
        # one thread: the full frame if it fits the budget, else a centre crop scaled by sample count
        est_1 = t_all * args.threads
        if est_1 <= args.one_thread_seconds * 1.5:
            t1, st1 = time_oracle(case, S, 1, warm=False)
            r["one_thread"] = {"ms_per_frame": 1e3 * t1, "frames_per_s": 1.0 / t1, "msamples_per_s": st1["samples"] / 1e6 / t1,
                               "window": "full frame"}
        else:
            side = int(case.W * (args.one_thread_seconds / est_1) ** 0.5) // 16 * 16
            side = max(32, min(case.W, side))
            x0 = (case.W - side) // 2
            t1, st1 = time_oracle(case, S, 1, window=(x0, x0, x0 + side, x0 + side), warm=False)
            frame_s = t1 * st["samples"] / max(1, st1["samples"])
            r["one_thread"] = {"ms_per_frame": 1e3 * frame_s, "frames_per_s": 1.0 / frame_s,
                               "msamples_per_s": st1["samples"] / 1e6 / t1,
                               "window": f"{side}x{side} centre crop ({t1:.1f} s), scaled to the frame by sample count"}
        if have_gpu:
            R = case.hip_renderer()
            R.updateFrameID(0)
            for _ in range(3):
                R.render()
            ms = []
            for _ in range(10):
                R.render()
                ms.append(R.stats()["kernel_ms"])
            _, gst = R.renderStats()
            R.close()
            r["mi355x"] = {"kernel_ms": float(np.mean(ms)), "frames_per_s_kernel": 1e3 / float(np.mean(ms)),
                           "msamples_per_s": gst["samples"] / 1e3 / float(np.mean(ms)), "samples": gst["samples"]}
        out[name] = r
        del S
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
