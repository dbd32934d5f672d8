#!/bin/bash
# round 4: what the general 64-bit address form costs on its own: C4 (fits 32-bit offsets) with option addr64 = 1 against the default
O=gpurun_out
for rep in 1 2; do for a in 0 1; do
  python bench.py --steps 20 --cpu-baseline off --pmc off --option addr64=$a > $O/r04_i_addr64_$a.$rep.json 2> $O/r04_i_addr64_$a.$rep.err || tail -3 $O/r04_i_addr64_$a.$rep.err
done; done
python bench.py --steps 10 --cpu-baseline off --pmc off --fields 3 --option addr64=1 > $O/r04_i_f3_addr64.json 2> $O/r04_i_f3_addr64.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_i_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, "%.3f ms" % d["roofline"]["kernel_ms"])
PY
