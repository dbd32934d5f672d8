#!/bin/bash
# round 4: timing of form 1 with the sample position, cell coordinate, gradient, shading dots and the "over" fused as well
O=gpurun_out
bash tools/ab_variants.sh run > /dev/null; cat $O/variants/results.txt
for v in fuse0 fuse1; do for cfg in "--config c3_gear" "--fields 3" "--camera closeup"; do
  EXA_HIP_LIB=$PWD/build/variants/libexa_hip_$v.so python bench.py --cpu-baseline off --pmc off --steps 10 $cfg > $O/r04_r_tmp.json 2>/dev/null && python -c "import json; d=json.loads(open('$O/r04_r_tmp.json').read().strip().splitlines()[-1]); print('$v $cfg : %.3f ms' % d['roofline']['kernel_ms'])"
done; done
