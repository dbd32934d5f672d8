#!/bin/bash
# round 4: the other configurations with the per-axis basis sums (and the source order beside it where it is quick)
set -o pipefail
O=gpurun_out
b() { name=$1; shift; python bench.py --cpu-baseline off --pmc off "$@" > $O/r04_h_$name.json 2> $O/r04_h_$name.err || tail -3 $O/r04_h_$name.err; }
for f in 1 0; do
b c2_f$f --config c2_lanl --size 1024 --steps 50 --basis-form $f
b c3iso_f$f --config c3_gear --iso 0.5 --steps 20 --basis-form $f
b c3_f$f --config c3_gear --steps 20 --basis-form $f
b f3_f$f --fields 3 --steps 10 --basis-form $f
b s125_f$f --scale 1.25 --steps 10 --basis-form $f
b closeup_f$f --camera closeup --steps 10 --basis-form $f
b c5_f$f --size 4096 --iso 0.5 --ao --spp 16 --steps 3 --warmup 1 --basis-form $f
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_h_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-34s fps %.3f ms %.3f Msamples/s %.0f kernel_ms %.3f" % (f.split('/')[-1], d["value"], d["ms_per_step"], d["msamples_per_s"], r["kernel_ms"]))
    except Exception as e: print(f, "ERR", e)
PY
