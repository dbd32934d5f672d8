#!/bin/bash
# round 5, last: counter summary, kernel trace and phase clocks of the final build (the three files r05_z.sh feeds to collect_profiles.py)
set -o pipefail
O=gpurun_out
stop() { rc=$1; if [ "$rc" -ge 124 ]; then echo "step killed (rc $rc): stopping"; exit "$rc"; fi; }
timeout -k 10 400 bash tools/config_timeline.sh $O/r05_z_tl_c4 --steps 10 --pmc off --in-flight 1 | cut -c1-220; stop $?
timeout -k 10 1000 bash tools/pmc_run.sh $O/r05_z_pmc_c4 --pmc off --in-flight 1 > $O/r05_z_pmc_c4.txt 2>&1; stop $?
grep -E "^==|SQ_INSTS_VALU |lane util|WAIT_ANY/|FETCH_SIZE .*GB|L2 hit|L1 miss" $O/r05_z_pmc_c4.txt | head -24
timeout -k 10 400 python tests/gpu_diag.py > $O/r05_z_diag.txt 2>&1; stop $?; tail -14 $O/r05_z_diag.txt
echo done
