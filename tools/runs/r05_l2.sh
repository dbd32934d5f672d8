#!/bin/bash
# round 5, late: cost classes x static tile order again, on the seven-wave march and with four frames in flight (the bench's protocol;
# r05_l ran one frame at a time): does the frame's tail still outweigh cache locality when the next frame covers it?
set -o pipefail
O=gpurun_out
stop() { rc=$1; if [ "$rc" -ge 124 ]; then echo "step killed (rc $rc): stopping"; exit "$rc"; fi; }
for to in 4 6; do
for cl in 1 8 32 256; do
  name=to${to}_cl${cl}
  EXA_COST_CLASSES=$cl timeout -k 10 300 python bench.py --cpu-baseline off --pmc off --steps 30 --tile-order $to > $O/r05_l2_$name.json 2> $O/r05_l2_$name.err; rc=$?; stop $rc; [ $rc -ne 0 ] && tail -3 $O/r05_l2_$name.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_l2_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-20s ms per frame (F=4) %.3f   one at a time %.3f   kernel %.3f" % (f.split('/')[-1], d["ms_per_step"], d["latency_ms"], r["kernel_ms"]))
    except Exception as e: print(f, "ERR", e)
PY
echo done
