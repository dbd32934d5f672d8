#!/bin/bash
# round 4: compact segment queue (8-byte slots, entry distance stored only when it does not follow from the previous exit)
set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_interleave.py -x -q > $O/r04_o_tests.log 2>&1; tail -3 $O/r04_o_tests.log
grep -q " passed" $O/r04_o_tests.log || exit 1
bash tools/ab_variants.sh run > /dev/null; cat $O/variants/results.txt
for v in cq0 cq1; do for cfg in "--config c3_gear --iso 0.5" "--fields 3" "--camera closeup"; do
  EXA_HIP_LIB=$PWD/build/variants/libexa_hip_$v.so python bench.py --cpu-baseline off --pmc off --steps 10 $cfg > $O/r04_o_tmp.json 2>/dev/null && python -c "import json; d=json.loads(open('$O/r04_o_tmp.json').read().strip().splitlines()[-1]); print('$v $cfg : %.3f ms' % d['roofline']['kernel_ms'])"
done; done
