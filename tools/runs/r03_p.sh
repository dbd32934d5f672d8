#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_interleave.py -x -q > $O/r03_p_tests.log 2>&1; tail -3 $O/r03_p_tests.log
(cd tests && python gpu_fuzz_sched.py 600 679 --keep-going > ../$O/r03_p_fuzz_sched.log 2>&1; tail -2 ../$O/r03_p_fuzz_sched.log)
for d in 0 1 4 8; do
  python bench.py --size 4096 --iso 0.5 --ao --spp 16 --steps 3 --warmup 1 --cpu-baseline off --pmc off --option ao_defer=$d > $O/r03_p_c5_defer$d.json 2> $O/r03_p_c5_defer$d.err
  python bench.py --config c3_gear --iso 0.5 --ao --steps 20 --cpu-baseline off --pmc off --option ao_defer=$d > $O/r03_p_c3isoao_defer$d.json 2> $O/r03_p_c3isoao_defer$d.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_p_c*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "%.3f ms/step  kernel %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"]))
PY
bash tools/config_timeline.sh $O/r03_p_tl_c5_coop4 --size 4096 --iso 0.5 --ao --spp 16 --steps 2 --warmup 1 --pmc off --option prepass_split=0 --option ao_defer=4
