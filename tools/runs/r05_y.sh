#!/bin/bash
# round 5, late: AO rays in front of the march again (ao_overlap 0 is the default beside the seven-wave march): the AO tests, C5 and
# C3 + iso + AO with either plan, the C5 line of r05_z and its kernel trace again
set -o pipefail
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "ao or c5 or C5 or surf" > $O/r05_y_tests.log 2>&1; rc=$?; tail -3 $O/r05_y_tests.log; [ $rc -ne 0 ] && exit $rc
for o in 0 1 0 1; do
  timeout -k 10 300 python bench.py --cpu-baseline off --pmc off --config c3_gear --iso 0.5 --ao --steps 20 --option ao_overlap=$o 2>$O/r05_y_c3ao_$o.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 + iso + AO, ao_overlap $o: %.3f ms per frame, one at a time %.3f' % (d['ms_per_step'], d['latency_ms']))" || exit 1
done
timeout -k 10 700 python bench.py --cpu-baseline off --pmc on --size 4096 --iso 0.5 --ao --spp 16 --steps 3 --warmup 1 > $O/r05_z_c5.json 2> $O/r05_z_c5.err; rc=$?; [ $rc -ge 124 ] && exit $rc
python -c "
import json; d=json.loads(open('gpurun_out/r05_z_c5.json').read().strip().splitlines()[-1]); r=d['roofline']; print('C5: %.1f ms per 16 samples, %.3f frames/s, one at a time %.1f, kernel %.2f, Msamples/s %.0f, valu %.3f' % (d['ms_per_step'], d['value'], d['latency_ms'], r['kernel_ms'], d['msamples_per_s'], r.get('frac') or 0)); print(r.get('per_kernel'))"
timeout -k 10 400 bash tools/config_timeline.sh $O/r05_y_tl_c5 --size 4096 --iso 0.5 --ao --spp 16 --steps 2 --warmup 1 --pmc off --in-flight 1 | cut -c1-200
echo done
