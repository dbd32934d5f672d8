#!/bin/bash
# round 5, late: no cost ordering (EXA_COST_CLASSES=1) against 256 classes, four frames in flight, on the other configurations and on rank 0 of 8
set -o pipefail
O=gpurun_out
stop() { rc=$1; if [ "$rc" -ge 124 ]; then echo "step killed (rc $rc): stopping"; exit "$rc"; fi; }
b() { name=$1; shift; for cl in 1 256; do EXA_COST_CLASSES=$cl timeout -k 10 400 python bench.py --cpu-baseline off --pmc off "$@" > $O/r05_l3_${name}_cl$cl.json 2> $O/r05_l3_${name}_cl$cl.err; rc=$?; stop $rc; [ $rc -ne 0 ] && tail -3 $O/r05_l3_${name}_cl$cl.err; done; }
b c2 --config c2_lanl --size 1024 --steps 100
b c3 --config c3_gear --steps 30
b c3iso --config c3_gear --iso 0.5 --steps 30
b closeup --camera closeup --steps 20
b s125 --scale 1.25 --steps 10
b c5 --size 4096 --iso 0.5 --ao --spp 16 --steps 3 --warmup 1
export EXA_BENCH_FORCE_DIST=1 EXA_BENCH_SHARD=0,8
b r0of8 --steps 60
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_l3_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-24s ms per frame (F=4) %.3f   one at a time %.3f   kernel %.3f" % (f.split('/')[-1], d["ms_per_step"], d["latency_ms"], r["kernel_ms"]))
    except Exception as e: print(f, "ERR", e)
PY
echo done
