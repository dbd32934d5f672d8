#!/bin/bash
# round 4: lanes of a wave along a Morton curve of their 8x8 pixel block (quad = 2x2 pixels): frame time and L1 accesses
O=gpurun_out
bash tools/ab_variants.sh run
cat $O/variants/results.txt
K="form1::renderFrameKdKernel<true, true, 0, false, 0, true, 0>"
for v in morton0 morton1; do
  EXA_HIP_LIB=$PWD/build/variants/libexa_hip_$v.so bash tools/pmc_extra.sh $O/r04_f_$v "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" --pmc off > $O/r04_f_$v.log 2>&1
  echo $v; grep -A4 "$K" $O/r04_f_$v/summary.txt | head -5
done
