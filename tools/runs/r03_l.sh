#!/bin/bash
set -o pipefail
O=gpurun_out
EXA_WIDE_BUDGET_GB=40 python tests/gpu_wide_probe.py 1.0 256 > $O/r03_l_wide_probe_256.txt 2>&1; head -5 $O/r03_l_wide_probe_256.txt
for w in 2.2 2.9; do for sp in 1.63 1.2; do echo "== wide_top 16 work $w speed2 $sp"; EXA_WIDE_WORK_TOP=$w EXA_WIDE_SPEED2=$sp EXA_HIP_VERBOSE=1 python tests/gpu_shard_scaling.py 1.0 4 16 2>&1 | grep -E "world 8|x16 lanes" | tail -3; done; done | tee $O/r03_l_shard_top16.txt
echo done
