#!/bin/bash
# one extra PMC pass with a custom counter list: tools/pmc_extra.sh <outdir> "<counters>" [bench args]
# (keep to two counters per hardware block — TA, TD, TCP ...: a larger request is refused by the profiler at the first
# dispatch and the profiled process then lingers until the timeout)
set -u
OUT=$1; CNT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
case "$OUT" in /*) ;; *) OUT="$PWD/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${PMC_TIMEOUT:-150} rocprofv3 --pmc $CNT --output-format csv -d "$OUT/pass0" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --cpu-baseline off "$@" > "$OUT/pass0.json" 2> "$OUT/pass0.log" || { grep -m2 "exceeds\|rror" "$OUT/pass0.log"; exit 1; }
python3 "$ROOT/tools/pmc_summary.py" "$OUT" | tee "$OUT/summary.txt"
