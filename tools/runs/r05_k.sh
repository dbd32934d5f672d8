#!/bin/bash
# round 5: fields beyond 4 GiB (scale 1.25): launch order of the tiles over the XCDs and memory order of the bricks against HBM traffic
set -o pipefail
O=gpurun_out
stop() { rc=$1; if [ "$rc" -ge 124 ]; then echo "step killed (rc $rc): stopping"; exit "$rc"; fi; }
b() { name=$1; shift; timeout -k 10 500 python bench.py --cpu-baseline off --in-flight 1 --scale 1.25 --steps 10 "$@" > $O/r05_k_$name.json 2> $O/r05_k_$name.err; rc=$?; stop $rc; [ $rc -ne 0 ] && tail -3 $O/r05_k_$name.err; }
b to4_bo0 --pmc on
b to4_bo1 --pmc on --option brick_order=1
b to5_bo0 --pmc on --tile-order 5
b to6_bo0 --pmc on --tile-order 6
b to7_bo0 --pmc on --tile-order 7
b to6_bo1 --pmc on --tile-order 6 --option brick_order=1
b to4_nofb --pmc on --option tile_feedback=0
b to6_nofb --pmc on --tile-order 6 --option tile_feedback=0
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_k_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-24s ms %.3f kernel %.3f traffic %.2f GB hbm %.3f valu %.3f" % (f.split('/')[-1], d["ms_per_step"], r["kernel_ms"], (r.get("traffic") or 0)/1e9, (r.get("hbm_measured") or {}).get("frac",0), r.get("frac") or 0))
    except Exception as e: print(f, "ERR", e)
PY
# scenes with empty cells (the form0e kernels; their rope variants spill 1-4 dwords): does the rope walk still win?
EXA_WALK_HOLES=0.05 timeout -k 10 600 python tests/gpu_walk_choice.py c4_exajet 2048 > $O/r05_k_walk_holes_c4.txt 2>&1; stop $?; head -3 $O/r05_k_walk_holes_c4.txt
EXA_WALK_HOLES=0.05 timeout -k 10 600 python tests/gpu_walk_choice.py c3_gear 2048 > $O/r05_k_walk_holes_c3.txt 2>&1; stop $?; head -3 $O/r05_k_walk_holes_c3.txt
echo done
