#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests/test_host_cli.py -x -q > $O/r03_o_tests.log 2>&1; tail -2 $O/r03_o_tests.log
EXA_BENCH_FORCE_DIST=1 EXA_BENCH_SHARD=0,8 python bench.py --steps 20 > $O/r03_o_rank0of8_nccl.json 2> $O/r03_o_rank0of8_nccl.err || tail -5 $O/r03_o_rank0of8_nccl.err
EXA_BENCH_FORCE_DIST=1 python bench.py --steps 10 --cpu-baseline off --pmc off > $O/r03_o_1rank_nccl.json 2> $O/r03_o_1rank_nccl.err || tail -5 $O/r03_o_1rank_nccl.err
python - <<'PY'
import json
for f in ("gpurun_out/r03_o_rank0of8_nccl.json","gpurun_out/r03_o_1rank_nccl.json"):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, d["value"], d["ms_per_step"], d["latency_ms"], d["frames_in_flight"], d.get("rehearsal_of"), r["frac"], r.get("pmc_source","")[:80], r.get("pmc_live_error"))
PY
