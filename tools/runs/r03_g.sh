#!/bin/bash
# round 3, GPU run G: deferred AO rays
set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_interleave.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_multi.py -x -q > $O/r03_g_tests.log 2>&1; tail -3 $O/r03_g_tests.log
(cd tests && python gpu_fuzz.py 3300 3419 --rich --keep-going > ../$O/r03_g_fuzz_rich.log 2>&1; tail -2 ../$O/r03_g_fuzz_rich.log; python gpu_fuzz_sched.py 200 279 --keep-going > ../$O/r03_g_fuzz_sched.log 2>&1; tail -2 ../$O/r03_g_fuzz_sched.log)
for d in 1 0; do
  python bench.py --size 4096 --iso 0.5 --ao --spp 16 --steps 3 --warmup 1 --cpu-baseline off --pmc off --option ao_defer=$d > $O/r03_g_c5_defer$d.json 2> $O/r03_g_c5_defer$d.err
  python bench.py --config c3_gear --iso 0.5 --ao --steps 20 --cpu-baseline off --pmc off --option ao_defer=$d > $O/r03_g_c3isoao_defer$d.json 2> $O/r03_g_c3isoao_defer$d.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_g_c*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "%.3f ms/step  kernel %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"]))
PY
bash tools/config_timeline.sh $O/r03_g_tl_c5 --size 4096 --iso 0.5 --ao --spp 16 --steps 2 --warmup 1 --pmc off --option prepass_split=0
echo done
