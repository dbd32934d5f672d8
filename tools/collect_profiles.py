#!/usr/bin/env python3
"""Copies the reference measurements of a round's GPU session (tools/runs/rNN_z.sh -> gpurun_out/rNN_z_*) into profiles/:
   python tools/collect_profiles.py r05"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
O, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


shutil.copy(os.path.join(O, f"{tag}_z_bench.json"), os.path.join(P, f"{tag}_final_bench.json"))
names = {"c2": "c2_lanl_1024", "c3iso": "c3_gear_2048_dvr_iso", "c3": "c3_gear_2048_dvr", "c5": "c5_exajet_4096_iso_ao_16spp",
         "closeup": "c4_closeup_camera", "f3": "c4_three_channels", "s125": "c4_scale125", "form0": "c4_basis_form0_source_order",
         "stack": "c4_stack_walk"}
other = {}
for k, n in names.items():
    f = os.path.join(O, f"{tag}_z_{k}.json")
    if os.path.exists(f):
        other[n] = last_json(f)
json.dump(other, open(os.path.join(P, f"{tag}_other_configs.json"), "w"), indent=1)
for src, dst in ((f"{tag}_z_pmc_c4/summary.txt", f"{tag}_final_pmc_summary.txt"), (f"{tag}_z_diag.txt", f"{tag}_phase_diag_c4.txt"),
                 (f"{tag}_z_shard_scaling.txt", f"{tag}_shard_scaling_rehearsal.txt"), (f"{tag}_z_rank0of8_nccl.json", f"{tag}_rank0_of_8_rehearsal_nccl.json")):
    if os.path.exists(os.path.join(O, src)):
        shutil.copy(os.path.join(O, src), os.path.join(P, dst))
stats = glob.glob(os.path.join(O, f"{tag}_z_tl_c4", "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:          # gpurun_out/ is merged across sessions: the newest trace is this session's
    shutil.copy(max(stats, key=os.path.getmtime), os.path.join(P, f"{tag}_kernel_stats.csv"))
print("copied", sorted(f for f in os.listdir(P) if f.startswith(tag)))
