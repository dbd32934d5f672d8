#!/bin/bash
# round 4: guarded first-sample correction loops + in-place packed-field updates (pop / push / accept): parity, then frame times
set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_interleave.py tests/test_gpu_empty_cells.py -x -q > $O/r04_q_tests.log 2>&1; tail -3 $O/r04_q_tests.log
grep -q " passed" $O/r04_q_tests.log || exit 1
for rep in 1 2; do python bench.py --cpu-baseline off --pmc off --steps 20 > $O/r04_q_tmp.json 2>/dev/null && python -c "import json; d=json.loads(open('$O/r04_q_tmp.json').read().strip().splitlines()[-1]); print('C4: %.3f ms' % d['roofline']['kernel_ms'])"; done
for cfg in "--config c3_gear --iso 0.5" "--config c3_gear" "--fields 3" "--camera closeup" "--config c2_lanl --size 1024 --steps 50"; do
  python bench.py --cpu-baseline off --pmc off --steps 10 $cfg > $O/r04_q_tmp.json 2>/dev/null && python -c "import json; d=json.loads(open('$O/r04_q_tmp.json').read().strip().splitlines()[-1]); print('$cfg : %.3f ms' % d['roofline']['kernel_ms'])"
done
