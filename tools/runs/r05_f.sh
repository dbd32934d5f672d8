#!/bin/bash
# round 5: AO rays beside the march (ao_overlap): scheduling sweep, AO parity cases, C5 / C3 + iso + AO timings with and without
set -o pipefail
O=gpurun_out
stop() { rc=$1; if [ "$rc" -ge 124 ]; then echo "step killed (rc $rc): stopping"; exit "$rc"; fi; }
( cd tests && timeout -k 10 600 python gpu_fuzz_sched.py 0 39 --keep-going ) > $O/r05_f_sched.log 2>&1; rc=$?; stop $rc; tail -3 $O/r05_f_sched.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "ao or iso or mesh or contour" > $O/r05_f_parity_surf.log 2>&1; rc=$?; stop $rc; tail -3 $O/r05_f_parity_surf.log
[ $rc -ne 0 ] && exit $rc
b() { name=$1; shift; timeout -k 10 500 python bench.py --cpu-baseline off --pmc off --in-flight 1 "$@" > $O/r05_f_$name.json 2> $O/r05_f_$name.err; rc=$?; stop $rc; [ $rc -ne 0 ] && tail -3 $O/r05_f_$name.err; }
for ov in 0 1; do
  b c5_ov$ov --size 4096 --iso 0.5 --ao --spp 16 --steps 3 --warmup 1 --option ao_overlap=$ov
  b c3isoao_ov$ov --config c3_gear --iso 0.5 --ao --steps 20 --option ao_overlap=$ov
done
b c3iso --config c3_gear --iso 0.5 --steps 20
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_f_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-28s ms %.3f kernel(last launch) %.3f" % (f.split('/')[-1], d["ms_per_step"], r["kernel_ms"]))
    except Exception as e: print(f, "ERR", e)
PY
echo done
