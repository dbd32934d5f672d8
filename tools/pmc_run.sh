#!/bin/bash
# Collects rocprofv3 PMC counters for bench.py in separate passes (one rocprofv3 run per
# counter group; never combined with trace options) and prints a per-kernel summary.
# usage (on the GPU box): tools/pmc_run.sh <outdir> [bench args...]
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
case "$OUT" in /*) ;; *) OUT="$PWD/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD"
 "FETCH_SIZE"
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum"
 "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT"
)
i=0
for p in "${PASSES[@]}"; do
  timeout -k 10 400 rocprofv3 --pmc $p --output-format csv -d "$OUT/pass$i" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --cpu-baseline off "$@" > "$OUT/pass$i.json" 2> "$OUT/pass$i.log" || { echo "pass $i failed"; tail -3 "$OUT/pass$i.log"; }
  i=$((i+1))
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" | tee "$OUT/summary.txt"
