#!/bin/bash
# round 5, late (seven-wave rope march): rank 0 of an N-GPU job for N = 1, 2, 4, 8 through the RCCL path (four frames in flight, one GPU),
# kernel traces of C5 and C3 + iso
set -o pipefail
O=gpurun_out
for w in 1 2 4 8; do
  EXA_BENCH_FORCE_DIST=1 EXA_BENCH_SHARD=0,$w timeout -k 10 400 python bench.py --steps 40 --cpu-baseline off --pmc off > $O/r05_x_r0of$w.json 2> $O/r05_x_r0of$w.err; rc=$?; [ $rc -ge 124 ] && exit $rc
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_x_r0of*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print("%-24s ms/frame %.3f latency %.3f fps %.2f F %d" % (f.split('/')[-1], d["ms_per_step"], d["latency_ms"], d["value"], d["frames_in_flight"]))
    except Exception as e: print(f, "ERR", e)
PY
timeout -k 10 400 bash tools/config_timeline.sh $O/r05_x_tl_c5 --size 4096 --iso 0.5 --ao --spp 16 --steps 2 --warmup 1 --pmc off --in-flight 1 | cut -c1-200; rc=$?; [ $rc -ge 124 ] && exit $rc
timeout -k 10 400 bash tools/config_timeline.sh $O/r05_x_tl_c3iso --config c3_gear --iso 0.5 --steps 10 --pmc off --in-flight 1 | cut -c1-200
echo done
