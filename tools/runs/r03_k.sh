#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "wide" > $O/r03_k_tests.log 2>&1; tail -3 $O/r03_k_tests.log
(cd tests && python gpu_fuzz_sched.py 500 539 --keep-going > ../$O/r03_k_fuzz_sched.log 2>&1; tail -2 ../$O/r03_k_fuzz_sched.log)
EXA_WIDE_BUDGET_GB=40 python tests/gpu_wide_probe.py > $O/r03_k_wide_probe.txt 2>&1; cat $O/r03_k_wide_probe.txt
echo done
