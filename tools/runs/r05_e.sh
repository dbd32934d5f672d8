#!/bin/bash
# round 5: the configuration tests with the new whole-frame checks (form 1 vs form 0, rope vs stack), which walk for which TF
set -o pipefail
O=gpurun_out
stop() { rc=$1; if [ "$rc" -ge 124 ]; then echo "step killed (rc $rc): stopping"; exit "$rc"; fi; }
timeout -k 10 1000 python -m pytest tests/test_gpu_configs.py -x -q -s > $O/r05_e_configs.log 2>&1; rc=$?; stop $rc; tail -3 $O/r05_e_configs.log; grep -E "^C[2345]|form|pixels beyond" $O/r05_e_configs.log | cut -c1-260
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tests/gpu_walk_choice.py c4_exajet 2048 > $O/r05_e_walk_choice_c4.txt 2>&1; stop $?; cat $O/r05_e_walk_choice_c4.txt | tail -8
timeout -k 10 300 python tests/gpu_walk_choice.py c2_lanl 1024 > $O/r05_e_walk_choice_c2.txt 2>&1; stop $?; cat $O/r05_e_walk_choice_c2.txt | tail -8
echo done
