#!/bin/bash
# per-kernel time of one bench configuration: rocprofv3 --kernel-trace --stats of bench.py <args>
# usage: tools/config_timeline.sh <outdir> <bench args...>
set -u
OUT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
case "$OUT" in /*) ;; *) OUT="$PWD/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --cpu-baseline off "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
find "$OUT/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -1 {}; grep -E "render|surface|aoRays|Activity|Refit" {}' | cut -c1-200
