#!/bin/bash
# round 4: form 1 with the fused sample position / cell coordinate / gradient / shading dots / "over", mirrored in the oracle: whole GPU suite, frame times
set -o pipefail
O=gpurun_out
python -m pytest tests -q -m gpu -x > $O/r04_s_gpu_suite.log 2>&1; tail -4 $O/r04_s_gpu_suite.log
grep -q " passed" $O/r04_s_gpu_suite.log || exit 1
for rep in 1 2; do python bench.py --cpu-baseline off --pmc off --steps 20 > $O/r04_s_tmp.json 2>/dev/null && python -c "import json; d=json.loads(open('$O/r04_s_tmp.json').read().strip().splitlines()[-1]); print('C4: %.3f ms' % d['roofline']['kernel_ms'])"; done
python bench.py --steps 20 --pmc off > $O/r04_s_bench_cpu.json 2>/dev/null; python -c "import json; d=json.loads(open('$O/r04_s_bench_cpu.json').read().strip().splitlines()[-1]); c=d['cpu_baseline']; print('crop vs oracle: max', c['crop_max_abs_diff_rgba8'], 'pixels', c['crop_pixels_differing'])"
