#!/bin/bash
# round 4: timing probe — two 16-byte cell loads per brick visit instead of four 8-byte ones (wrong pixels; what a row-pair layout would issue)
O=gpurun_out
bash tools/ab_variants.sh run > /dev/null
cat $O/variants/results.txt
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/variants/*.1.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["config"]["samples_per_frame"], d["roofline"]["kernel_ms"])
PY
K="form1::renderFrameKdKernel<true, true, 0, false, 0, true, 0>"
for v in two; do
  EXA_HIP_LIB=$PWD/build/variants/libexa_hip_$v.so bash tools/pmc_extra.sh $O/r04_g_$v "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" --pmc off > $O/r04_g_$v.log 2>&1
  echo $v; grep -A4 "$K" $O/r04_g_$v/summary.txt | head -5
done
