#!/bin/bash
# round 5, reference measurements of the committed state (the copies cited in DESIGN.md are under profiles/r05_*)
set -o pipefail
O=gpurun_out
stop() { rc=$1; if [ "$rc" -ge 124 ]; then echo "step killed (rc $rc): stopping"; exit "$rc"; fi; }
timeout -k 10 900 python bench.py > $O/r05_z_bench.json 2> $O/r05_z_bench.err; stop $?; tail -c 600 $O/r05_z_bench.json; echo
timeout -k 10 400 bash tools/config_timeline.sh $O/r05_z_tl_c4 --steps 10 --pmc off --in-flight 1 | cut -c1-220; stop $?
timeout -k 10 1100 bash tools/pmc_run.sh $O/r05_z_pmc_c4 --pmc off --in-flight 1 > $O/r05_z_pmc_c4.txt 2>&1; stop $?
grep -E "^==|SQ_INSTS_VALU |lane util|WAVE_CYCLES|FETCH_SIZE .*GB|L2 hit|L1 miss" $O/r05_z_pmc_c4.txt | head -24
b() { name=$1; shift; timeout -k 10 700 python bench.py --cpu-baseline off --pmc on "$@" > $O/r05_z_$name.json 2> $O/r05_z_$name.err; rc=$?; stop $rc; [ $rc -ne 0 ] && tail -3 $O/r05_z_$name.err; }
b c2 --config c2_lanl --size 1024 --steps 50
b c3iso --config c3_gear --iso 0.5 --steps 20
b c3 --config c3_gear --steps 20
b f3 --fields 3 --steps 10
b s125 --scale 1.25 --steps 10
b closeup --camera closeup --steps 10
b c5 --size 4096 --iso 0.5 --ao --spp 16 --steps 3 --warmup 1
b form0 --basis-form 0 --steps 20
b stack --option walk=1 --steps 20
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05_z_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print("%-30s fps %.3f ms %.3f lat %.3f Msamples/s %.0f valu %.3f hbm %.3f traffic %.2f GB kernel %.3f" % (f.split('/')[-1], d["value"], d["ms_per_step"], d["latency_ms"], d["msamples_per_s"], r.get("frac") or 0, (r.get("hbm_measured") or {}).get("frac",0), (r.get("traffic") or 0)/1e9, r["kernel_ms"]))
    except Exception as e: print(f, "ERR", e)
PY
timeout -k 10 400 python tests/gpu_diag.py > $O/r05_z_diag.txt 2>&1; stop $?; tail -14 $O/r05_z_diag.txt
timeout -k 10 400 python tests/gpu_shard_scaling.py 1.0 4 > $O/r05_z_shard_scaling.txt 2>&1; stop $?; tail -8 $O/r05_z_shard_scaling.txt
EXA_BENCH_FORCE_DIST=1 EXA_BENCH_SHARD=0,8 timeout -k 10 400 python bench.py --steps 20 --cpu-baseline off > $O/r05_z_rank0of8_nccl.json 2> $O/r05_z_rank0of8_nccl.err || tail -5 $O/r05_z_rank0of8_nccl.err
python -c "
import json; d=json.loads(open('gpurun_out/r05_z_rank0of8_nccl.json').read().strip().splitlines()[-1]); print('rank 0 of 8 rehearsal: %.3f ms/frame (F=%d), latency %.3f ms' % (d['ms_per_step'], d['frames_in_flight'], d['latency_ms']))"
echo done
