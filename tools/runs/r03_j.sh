#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "wide or launch_order or scheduling" > $O/r03_j_tests.log 2>&1; tail -3 $O/r03_j_tests.log
(cd tests && python gpu_fuzz_sched.py 400 459 --keep-going > ../$O/r03_j_fuzz_sched.log 2>&1; tail -2 ../$O/r03_j_fuzz_sched.log)
EXA_WIDE_BUDGET_GB=40 python tests/gpu_wide_probe.py > $O/r03_j_wide_probe.txt 2>&1; cat $O/r03_j_wide_probe.txt
for top in 4 8 16; do echo "== wide_top $top"; python tests/gpu_shard_scaling.py 1.0 4 $top 2>&1 | tee -a $O/r03_j_shard_scaling_top$top.txt; done
echo done
