#!/bin/bash
# round 5, late: the seven-wave rope march (in-tree build) against the six-wave one of the previous commit and a build with the default
# machine scheduler (build/variants: tools/ab_variants.sh), then the parity tests and the configurations that take that kernel
set -o pipefail
O=gpurun_out
stop() { rc=$1; if [ "$rc" -ge 124 ]; then echo "step killed (rc $rc): stopping"; exit "$rc"; fi; }
tools/ab_variants.sh run; stop $?
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r05_u_gpu_tests.log 2>&1; rc=$?; tail -4 $O/r05_u_gpu_tests.log; stop $rc
[ $rc -ne 0 ] && exit $rc
b() { name=$1; shift; timeout -k 10 400 python bench.py --cpu-baseline off --pmc off "$@" > $O/r05_u_$name.json 2> $O/r05_u_$name.err; rc=$?; stop $rc; [ $rc -ne 0 ] && tail -3 $O/r05_u_$name.err; }
b c4 --steps 50
b c2 --config c2_lanl --size 1024 --steps 50
b s125 --scale 1.25 --steps 10
b closeup --camera closeup --steps 10
b c3 --config c3_gear --steps 20
b c3iso --config c3_gear --iso 0.5 --steps 20
b form0 --basis-form 0 --steps 20
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_u_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-30s fps %.3f ms %.3f lat %.3f kernel %.3f" % (f.split('/')[-1], d["value"], d["ms_per_step"], d["latency_ms"], r["kernel_ms"]))
    except Exception as e: print(f, "ERR", e)
PY
echo done
