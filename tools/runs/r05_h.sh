#!/bin/bash
# round 5: two samples of a one-brick segment per iteration (EXA_MARCH_PAIR) at 5 and 4 waves per SIMD against the shipped 6 x 1
set -o pipefail
O=gpurun_out
stop() { rc=$1; if [ "$rc" -ge 124 ]; then echo "step killed (rc $rc): stopping"; exit "$rc"; fi; }
for v in pair5 pair4; do
  EXA_HIP_LIB=$PWD/build/variants/libexa_hip_$v.so timeout -k 10 420 python tests/gpu_rope_quick.py > $O/r05_h_quick_$v.log 2>&1; rc=$?; stop $rc
  echo "$v: $(grep -c 'ok=True' $O/r05_h_quick_$v.log) ok"; grep -E "ok=False|: False|FAILURES|Error|error" $O/r05_h_quick_$v.log | head -5
  [ $rc -ne 0 ] && exit $rc
done
timeout -k 10 900 bash tools/ab_variants.sh run > $O/r05_h_ab.txt 2>&1; stop $?; cat $O/r05_h_ab.txt | tail -14
EXA_HIP_LIB=$PWD/build/variants/libexa_hip_pair5.so timeout -k 10 900 bash tools/pmc_run.sh $O/r05_h_pmc_pair5 --pmc off --in-flight 1 > $O/r05_h_pmc_pair5.txt 2>&1; stop $?
grep -E "^==|SQ_INSTS_VALU |lane util|WAVE_CYCLES|FETCH_SIZE .*GB|L2 hit|L1 miss" $O/r05_h_pmc_pair5.txt | head -14
echo done
