#!/bin/bash
# round 5: throughput of rank 0 of an N-GPU job for N = 2, 4, 8 (four frames in flight, RCCL path, one GPU); the default bench line once more
set -o pipefail
O=gpurun_out
for w in 2 4 8; do
  EXA_BENCH_FORCE_DIST=1 EXA_BENCH_SHARD=0,$w timeout -k 10 400 python bench.py --steps 40 --cpu-baseline off --pmc off > $O/r05_p_r0of$w.json 2> $O/r05_p_r0of$w.err; rc=$?; [ $rc -ge 124 ] && exit $rc
done
( time timeout -k 10 900 python bench.py > $O/r05_p_bench.json 2> $O/r05_p_bench.err ) 2> $O/r05_p_bench.time; rc=$?; [ $rc -ge 124 ] && exit $rc
grep real $O/r05_p_bench.time
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_p_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print("%-24s ms/frame %.3f latency %.3f fps %.2f F %d %s" % (f.split('/')[-1], d["ms_per_step"], d["latency_ms"], d["value"], d["frames_in_flight"], "note" if "ms_per_step_note" in d else ""))
    except Exception as e: print(f, "ERR", e)
PY
echo done
