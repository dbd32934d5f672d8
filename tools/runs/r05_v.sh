#!/bin/bash
# round 5, late: seven waves also for the rope march of frames WITH surfaces (build/variants: surf6 = the in-tree build, surf7 =
# -DEXA_ROPE_SURF_WAVES=7): C5 (iso + AO, 4096^2, 16 samples) and C4 + one iso-surface, twice, interleaved
set -o pipefail
O=gpurun_out; mkdir -p $O/v
: > $O/v/results.txt
for rep in 1 2; do
  for so in build/variants/libexa_hip_surf*.so; do
    name=$(basename "$so" .so); name=${name#libexa_hip_}
    EXA_HIP_LIB="$so" timeout -k 10 300 python bench.py --cpu-baseline off --pmc off --size 4096 --iso 0.5 --ao --spp 16 --steps 3 --warmup 1 > $O/v/c5_$name.$rep.json 2> $O/v/c5_$name.$rep.err
    rc=$?; [ $rc -ge 124 ] && { echo "killed"; exit $rc; }
    EXA_HIP_LIB="$so" timeout -k 10 300 python bench.py --cpu-baseline off --pmc off --iso 0.5 --steps 20 > $O/v/c4iso_$name.$rep.json 2> $O/v/c4iso_$name.$rep.err
    rc=$?; [ $rc -ge 124 ] && { echo "killed"; exit $rc; }
    python - "$name" "$rep" <<'PY' | tee -a gpurun_out/v/results.txt
import json,sys
n,r=sys.argv[1:3]
for c in ("c5","c4iso"):
    try:
        d=json.loads(open(f"gpurun_out/v/{c}_{n}.{r}.json").read().strip().splitlines()[-1])
        print("%-6s %-6s rep %s  %.3f ms/step  kernel %.3f  latency %.3f" % (c,n,r,d["ms_per_step"],d["roofline"]["kernel_ms"],d.get("latency_ms") or 0))
    except Exception as e: print(c,n,r,"ERR",e)
PY
  done
done
