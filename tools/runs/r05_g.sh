#!/bin/bash
# round 5: full GPU suite on the current state; PMC accounting of a frame with surfaces (raw lists kept for the test fixture)
set -o pipefail
O=gpurun_out
stop() { rc=$1; if [ "$rc" -ge 124 ]; then echo "step killed (rc $rc): stopping"; exit "$rc"; fi; }
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r05_g_gpu_suite.log 2>&1; rc=$?; stop $rc; tail -4 $O/r05_g_gpu_suite.log
[ $rc -ne 0 ] && exit $rc
mkdir -p $O/r05_g_csv_c3iso $O/r05_g_csv_c3
EXA_BENCH_KEEP_PMC_CSV=$PWD/$O/r05_g_csv_c3iso timeout -k 10 600 python bench.py --config c3_gear --iso 0.5 --steps 20 --cpu-baseline off --pmc on --in-flight 1 > $O/r05_g_c3iso.json 2> $O/r05_g_c3iso.err; stop $?
EXA_BENCH_KEEP_PMC_CSV=$PWD/$O/r05_g_csv_c3 timeout -k 10 600 python bench.py --config c3_gear --steps 20 --cpu-baseline off --pmc on --in-flight 1 > $O/r05_g_c3.json 2> $O/r05_g_c3.err; stop $?
python - <<'PY'
import json
for n in ("c3iso", "c3"):
    try:
        d=json.loads(open(f"gpurun_out/r05_g_{n}.json").read().strip().splitlines()[-1]); r=d["roofline"]
        print(n, "ms %.3f" % d["ms_per_step"], "frac", r.get("frac"), "valu", r.get("valu_wave_instructions_per_launch"), r.get("pmc_source","")[:140], r.get("pmc_live_error"))
        print("   per_kernel", json.dumps(r.get("per_kernel")))
    except Exception as e: print(n, "ERR", e)
PY
ls -la $O/r05_g_csv_c3iso | head
echo done
