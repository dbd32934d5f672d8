#!/bin/bash
# round 5: rope walk with 16-byte queue entries (first sample worked out at the leaf); both walks on every configuration
set -o pipefail
O=gpurun_out
stop() { rc=$1; if [ "$rc" -ge 124 ]; then echo "step killed (rc $rc): stopping"; exit "$rc"; fi; }
timeout -k 10 420 python tests/gpu_rope_quick.py > $O/r05_c_quick.log 2>&1; rc=$?; stop $rc; grep -c "ok=True" $O/r05_c_quick.log; grep -E "ok=False|: False|FAILURES|Error|error" $O/r05_c_quick.log | head -20
[ $rc -ne 0 ] && exit $rc
b() { name=$1; shift; timeout -k 10 400 python bench.py --cpu-baseline off --pmc off --in-flight 1 "$@" > $O/r05_c_$name.json 2> $O/r05_c_$name.err; rc=$?; stop $rc; [ $rc -ne 0 ] && tail -3 $O/r05_c_$name.err; }
for w in 1 2; do
  b c4_w$w --steps 20 --option walk=$w
  b c2_w$w --config c2_lanl --size 1024 --steps 50 --option walk=$w
  b c3_w$w --config c3_gear --steps 20 --option walk=$w
  b c3iso_w$w --config c3_gear --iso 0.5 --steps 20 --option walk=$w
  b f3_w$w --fields 3 --steps 10 --option walk=$w
  b closeup_w$w --camera closeup --steps 10 --option walk=$w
  b s125_w$w --scale 1.25 --steps 10 --option walk=$w
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_c_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-28s ms %.3f kernel %.3f form0 %s" % (f.split('/')[-1], d["ms_per_step"], r["kernel_ms"], r.get("kernel_ms_basis_form0")))
    except Exception as e: print(f, "ERR", e)
PY
echo done
