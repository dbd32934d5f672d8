#!/bin/bash
# round 4, first GPU session: the per-axis association of the basis sums (basis_form = 1) against the oracle in the same
# form, and an A/B of the two forms on the bench frame
set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "per_axis" > $O/r04_a_tests.log 2>&1; tail -3 $O/r04_a_tests.log
for rep in 1 2; do for f in 0 1; do
  python bench.py --steps 20 --cpu-baseline off --pmc off --basis-form $f > $O/r04_a_bench_f$f.$rep.json 2> $O/r04_a_bench_f$f.$rep.err || tail -5 $O/r04_a_bench_f$f.$rep.err
done; done
python bench.py --steps 20 --pmc off --basis-form 1 > $O/r04_a_bench_f1_cpu.json 2> $O/r04_a_bench_f1_cpu.err || tail -5 $O/r04_a_bench_f1_cpu.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04_a_bench_f*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, "%.2f fps %.3f ms kernel %.3f" % (d["value"], d["ms_per_step"], r["kernel_ms"]), d.get("cpu_baseline",{}).get("crop_max_abs_diff_rgba8"), d.get("cpu_baseline",{}).get("crop_pixels_differing"))
    except Exception as e: print(f, "failed", e)
PY
