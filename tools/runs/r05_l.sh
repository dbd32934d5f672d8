#!/bin/bash
# round 5: number of cost classes of the launch-order feedback x static tile order: frame tail against cache locality
set -o pipefail
O=gpurun_out
stop() { rc=$1; if [ "$rc" -ge 124 ]; then echo "step killed (rc $rc): stopping"; exit "$rc"; fi; }
for sc in 1.0 1.25; do
for to in 4 6; do
for cl in 1 4 16 64 256; do
  name=s${sc}_to${to}_cl${cl}
  EXA_COST_CLASSES=$cl timeout -k 10 400 python bench.py --cpu-baseline off --pmc off --in-flight 1 --scale $sc --steps 10 --tile-order $to > $O/r05_l_$name.json 2> $O/r05_l_$name.err; rc=$?; stop $rc; [ $rc -ne 0 ] && tail -3 $O/r05_l_$name.err
done; done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_l_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-28s ms %.3f kernel %.3f" % (f.split('/')[-1], d["ms_per_step"], r["kernel_ms"]))
    except Exception as e: print(f, "ERR", e)
PY
echo done
