#!/bin/bash
# round 5: frames in flight (one protocol for every N): rank 0 of 8 rehearsed through the RCCL path, and the whole frame on one GPU
set -o pipefail
O=gpurun_out
stop() { rc=$1; if [ "$rc" -ge 124 ]; then echo "step killed (rc $rc): stopping"; exit "$rc"; fi; }
for F in 1 2 3 4 6; do
  EXA_BENCH_FORCE_DIST=1 EXA_BENCH_SHARD=0,8 timeout -k 10 400 python bench.py --steps 40 --cpu-baseline off --pmc off --in-flight $F > $O/r05_m_r0of8_F$F.json 2> $O/r05_m_r0of8_F$F.err; stop $?
  timeout -k 10 400 python bench.py --steps 30 --cpu-baseline off --pmc off --in-flight $F > $O/r05_m_n1_F$F.json 2> $O/r05_m_n1_F$F.err; stop $?
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_m_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print("%-28s ms/frame %.3f latency %.3f fps %.2f" % (f.split('/')[-1], d["ms_per_step"], d["latency_ms"], d["value"]))
    except Exception as e: print(f, "ERR", e)
PY
timeout -k 10 600 bash tools/rehearse_ranks.sh $O/r05_m_reh 4 0.5; stop $?
echo done
