#!/bin/bash
# round 5, first look at the rope walk: the short division, parity of both walks on the quick cases, C4 with either walk
set -o pipefail
O=gpurun_out
stop() { rc=$1; if [ "$rc" -ge 124 ]; then echo "step killed (rc $rc): stopping"; exit "$rc"; fi; }
timeout -k 10 120 tools/micro/div_exact 4000 > $O/r05_a_div.log 2>&1; stop $?; cat $O/r05_a_div.log
timeout -k 10 420 python tests/gpu_rope_quick.py > $O/r05_a_quick.log 2>&1; stop $?; grep -c "ok=True" $O/r05_a_quick.log; grep -E "ok=False|: False|FAILURES|Error|error" $O/r05_a_quick.log | head -20
for w in 1 2; do
  EXA_HIP_VERBOSE=1 timeout -k 10 300 python bench.py --steps 20 --cpu-baseline off --pmc off --in-flight 1 --option walk=$w > $O/r05_a_w$w.json 2> $O/r05_a_w$w.err; stop $?
  python - <<PY
import json
try:
    d=json.loads(open("$O/r05_a_w$w.json").read().strip().splitlines()[-1]); print("walk $w: ms_per_step %.3f kernel_ms %.3f form0 %s" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"].get("kernel_ms_basis_form0")))
except Exception as e: print("walk $w: ERR", e)
PY
  grep -E "rope walk|stats:" $O/r05_a_w$w.err | cut -c1-400
done
echo done
