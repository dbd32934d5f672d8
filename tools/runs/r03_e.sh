#!/bin/bash
# round 3, GPU run E: pre-pass header cache + split launch plan
set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_interleave.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q > $O/r03_e_tests.log 2>&1; tail -3 $O/r03_e_tests.log
for sp in 1 0; do
  EXA_HIP_VERBOSE=1 python bench.py --config c3_gear --iso 0.5 --steps 20 --cpu-baseline off --pmc off --option prepass_split=$sp > $O/r03_e_c3iso_sp$sp.json 2> $O/r03_e_c3iso_sp$sp.err; grep -h "pre-pass costs" $O/r03_e_c3iso_sp$sp.err | tail -1
  EXA_HIP_VERBOSE=1 python bench.py --size 4096 --iso 0.5 --ao --spp 16 --steps 3 --warmup 1 --cpu-baseline off --pmc off --option prepass_split=$sp > $O/r03_e_c5_sp$sp.json 2> $O/r03_e_c5_sp$sp.err; grep -h "pre-pass costs" $O/r03_e_c5_sp$sp.err | tail -1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_e_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "%.3f ms/step  kernel %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"]))
PY
bash tools/config_timeline.sh $O/r03_e_tl_c3iso --config c3_gear --iso 0.5 --steps 10 --pmc off --option prepass_split=0
bash tools/config_timeline.sh $O/r03_e_tl_c5 --size 4096 --iso 0.5 --ao --spp 16 --steps 2 --warmup 1 --pmc off --option prepass_split=0
echo done
