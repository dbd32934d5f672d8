#!/bin/bash
# kernel timeline of one rank of an N-GPU job rehearsed on one GPU: rocprofv3 --kernel-trace of bench.py with
# EXA_BENCH_SHARD=0,N; prints start/end of every kernel of the last frames relative to the frame start
# usage: tools/shard_timeline.sh <outdir> [N]
set -u
OUT=$1; N=${2:-8}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
case "$OUT" in /*) ;; *) OUT="$PWD/$OUT";; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
EXA_BENCH_FORCE_DIST=1 EXA_BENCH_PIPELINE=0 EXA_BENCH_SHARD=0,$N timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --cpu-baseline off > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 - "$OUT" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "renderFrame" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# group into frames: a gap of more than 200 us between a kernel's start and the previous kernels' end starts a new frame
frames, cur, last_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and s > last_end + 200000 and cur:
        frames.append(cur); cur = []
    cur.append(r); last_end = max(last_end or 0, e)
if cur:
    frames.append(cur)
for fr in frames[-3:]:
    t0 = min(int(r["Start_Timestamp"]) for r in fr)
    print("frame:")
    for r in fr:
        name = r["Kernel_Name"].split("(")[0].replace("void exa::", "")
        print(f"   {(int(r['Start_Timestamp']) - t0) / 1e6:7.3f} -> {(int(r['End_Timestamp']) - t0) / 1e6:7.3f} ms  grid {r.get('Grid_Size', r.get('Grid_Size_X', '?')):>8}  {name[:70]}")
PY
