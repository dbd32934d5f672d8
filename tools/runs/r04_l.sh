#!/bin/bash
# round 4: deferred AO rays sorted by pixel block and direction class (ao_defer = 2): parity, then kernel times on C5 and C3 + AO
set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_interleave.py -x -q > $O/r04_l_tests.log 2>&1; tail -3 $O/r04_l_tests.log
grep -q passed $O/r04_l_tests.log || exit 1
echo "== C5 ao_defer 2"
bash tools/config_timeline.sh $O/r04_l_c5_defer2 --size 4096 --iso 0.5 --ao --spp 16 --steps 2 --warmup 1 --pmc off --option prepass_split=0 --option ao_defer=2 | cut -c1-200
echo "== C3 + iso + AO ao_defer 2"
bash tools/config_timeline.sh $O/r04_l_c3isoao2 --config c3_gear --iso 0.5 --ao --steps 10 --pmc off --option prepass_split=0 --option ao_defer=2 | cut -c1-200
for d in 0 1 2; do python bench.py --size 4096 --iso 0.5 --ao --spp 16 --steps 3 --warmup 1 --cpu-baseline off --pmc off --option ao_defer=$d > $O/r04_l_c5_d$d.json 2>/dev/null; python -c "import json; d=json.loads(open('$O/r04_l_c5_d$d.json').read().strip().splitlines()[-1]); print('C5 ao_defer $d: %.1f ms per 16-spp frame' % d['ms_per_step'])"; done
for d in 0 1 2; do python bench.py --config c3_gear --iso 0.5 --ao --steps 20 --cpu-baseline off --pmc off --option ao_defer=$d > $O/r04_l_c3_d$d.json 2>/dev/null; python -c "import json; d=json.loads(open('$O/r04_l_c3_d$d.json').read().strip().splitlines()[-1]); print('C3+iso+AO ao_defer $d: %.2f ms' % d['ms_per_step'])"; done
