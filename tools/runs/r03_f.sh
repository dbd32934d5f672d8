#!/bin/bash
# round 3, GPU run F: seeded sweeps over the new paths, PMC summaries for C3 / C5 pre-pass, builder-made bricks at full scale
set -o pipefail
O=gpurun_out
cd tests
python gpu_fuzz_sched.py 0 119 --keep-going > ../$O/r03_f_fuzz_sched.log 2>&1; tail -2 ../$O/r03_f_fuzz_sched.log
python gpu_fuzz.py 0 199 --many --keep-going > ../$O/r03_f_fuzz_many.log 2>&1; tail -2 ../$O/r03_f_fuzz_many.log
python gpu_fuzz.py 3300 3499 --rich --keep-going > ../$O/r03_f_fuzz_rich.log 2>&1; tail -2 ../$O/r03_f_fuzz_rich.log
python gpu_fuzz.py 1000 1199 --keep-going > ../$O/r03_f_fuzz_plain.log 2>&1; tail -2 ../$O/r03_f_fuzz_plain.log
cd ..
python bench.py --config c3_gear --iso 0.5 --steps 20 --cpu-baseline off --pmc on > $O/r03_f_c3iso.json 2> $O/r03_f_c3iso.err
python bench.py --size 4096 --iso 0.5 --ao --spp 16 --steps 3 --warmup 1 --cpu-baseline off --pmc on > $O/r03_f_c5.json 2> $O/r03_f_c5.err
python bench.py --config c2_lanl --size 1024 --steps 50 --cpu-baseline off --pmc on > $O/r03_f_c2.json 2> $O/r03_f_c2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03_f_c*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, "%.3f ms/step kernel %.3f valu frac %s hbm %s" % (d["ms_per_step"], r["kernel_ms"], r.get("frac"), (r.get("hbm_measured") or {}).get("frac")), r.get("per_kernel"))
PY
timeout -k 10 1000 python tools/builder_bench.py --scale 1.0 --frames 12 > $O/r03_f_builder_bricks_full.json 2> $O/r03_f_builder_bricks_full.err; tail -3 $O/r03_f_builder_bricks_full.err; cat $O/r03_f_builder_bricks_full.json | head -50
echo done
