#!/bin/bash
# round 4: the deferred AO kernel at 6 waves per SIMD (80 VGPRs): kernel times and frame times
O=gpurun_out
python -m pytest tests/test_gpu_interleave.py -x -q > $O/r04_m_tests.log 2>&1; tail -2 $O/r04_m_tests.log
bash tools/config_timeline.sh $O/r04_m_c5 --size 4096 --iso 0.5 --ao --spp 16 --steps 2 --warmup 1 --pmc off --option prepass_split=0 | grep -E "aoRays|surfacePrepassKdKernel<0" | cut -c1-160
bash tools/config_timeline.sh $O/r04_m_c3 --config c3_gear --iso 0.5 --ao --steps 10 --pmc off --option prepass_split=0 | grep -E "aoRays|surfacePrepassKdKernel<0" | cut -c1-160
for d in 0 1; do python bench.py --size 4096 --iso 0.5 --ao --spp 16 --steps 3 --warmup 1 --cpu-baseline off --pmc off --option ao_defer=$d > $O/r04_m_c5_d$d.json 2>/dev/null; python -c "import json; d=json.loads(open('$O/r04_m_c5_d$d.json').read().strip().splitlines()[-1]); print('C5 ao_defer $d: %.1f ms per 16-spp frame' % d['ms_per_step'])"; done
for d in 0 1; do python bench.py --config c3_gear --iso 0.5 --ao --steps 20 --cpu-baseline off --pmc off --option ao_defer=$d > $O/r04_m_c3_d$d.json 2>/dev/null; python -c "import json; d=json.loads(open('$O/r04_m_c3_d$d.json').read().strip().splitlines()[-1]); print('C3+iso+AO ao_defer $d: %.2f ms' % d['ms_per_step'])"; done
