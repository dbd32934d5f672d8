#!/bin/bash
# Rehearsal of `bench.py --gpus N` on a ONE-GPU box: N self-spawned ranks share device 0 and gather through gloo
# (the measured configuration is nccl = RCCL, one GPU per rank).  Checks that the N-rank frame equals the 1-rank
# frame byte for byte.  usage: tools/rehearse_ranks.sh <outdir> [N] [scale]
set -u
OUT=$1; N=${2:-4}; SCALE=${3:-0.5}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
cd "$ROOT"
python3 bench.py --gpus 1 --scale $SCALE --steps 3 --warmup 1 --cpu-baseline off --dump "$OUT/rank1.png" > "$OUT/rank1.json" 2> "$OUT/rank1.err" || { echo "1-rank run failed"; tail -5 "$OUT/rank1.err"; exit 1; }
EXA_BENCH_BACKEND=gloo EXA_BENCH_ONE_DEVICE=1 python3 bench.py --gpus $N --scale $SCALE --steps 3 --warmup 1 --cpu-baseline off --dump "$OUT/rank$N.png" > "$OUT/rank$N.json" 2> "$OUT/rank$N.err" || { echo "$N-rank run failed"; tail -5 "$OUT/rank$N.err"; exit 1; }
python3 - "$OUT" $N <<'PY'
import json, sys
out, n = sys.argv[1], int(sys.argv[2])
a, b = open(f"{out}/rank1.png", "rb").read(), open(f"{out}/rank{n}.png", "rb").read()
j1, jn = json.loads(open(f"{out}/rank1.json").read().strip().splitlines()[-1]), json.loads(open(f"{out}/rank{n}.json").read().strip().splitlines()[-1])
print(f"1 rank : n_gpus {j1['n_gpus']} n_ranks_seen {j1['n_ranks_seen']} {j1['value']:.2f} fps  tiling: {j1['config']['tiling']}")
print(f"{n} ranks: n_gpus {jn['n_gpus']} n_ranks_seen {jn['n_ranks_seen']} {jn['value']:.2f} fps  tiling: {jn['config']['tiling']}")
print("frames identical byte for byte:", a == b, f"({len(a)} bytes of PNG)")
sys.exit(0 if (a == b and jn["n_gpus"] == n and jn["n_ranks_seen"] == n) else 1)
PY
