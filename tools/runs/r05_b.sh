#!/bin/bash
# round 5: the GPU suite with the rope walk in the matrix, then where the rope march spends its time
set -o pipefail
O=gpurun_out
stop() { rc=$1; if [ "$rc" -ge 124 ]; then echo "step killed (rc $rc): stopping"; exit "$rc"; fi; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r05_b_gpu_suite.log 2>&1; rc=$?; stop $rc; tail -5 $O/r05_b_gpu_suite.log
[ $rc -ne 0 ] && exit $rc
for w in 1 2; do
  EXA_DIAG_WALK=$w timeout -k 10 300 python tests/gpu_diag.py > $O/r05_b_diag_w$w.txt 2>&1; stop $?; tail -16 $O/r05_b_diag_w$w.txt
done
timeout -k 10 900 bash tools/pmc_run.sh $O/r05_b_pmc_rope --pmc off --in-flight 1 --option walk=2 > $O/r05_b_pmc_rope.txt 2>&1; stop $?
grep -E "^==|SQ_INSTS_VALU |lane util|WAVE_CYCLES|FETCH_SIZE .*GB|L2 hit|L1 miss|SQ_INSTS_V" $O/r05_b_pmc_rope.txt | head -40
echo done
