#!/bin/bash
# round 4: where the wave time goes in either form of the basis sums (phase clocks + PMC), after the NaN-position count fix
set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "per_axis and ex4_iso_noshade" > $O/r04_b_tests.log 2>&1; tail -2 $O/r04_b_tests.log
for f in 0 1; do EXA_BASIS_FORM=$f python tests/gpu_diag.py > $O/r04_b_diag_f$f.txt 2>&1; tail -12 $O/r04_b_diag_f$f.txt; done
for f in 0 1; do bash tools/pmc_run.sh $O/r04_b_pmc_f$f --basis-form $f --pmc off > $O/r04_b_pmc_f$f.log 2>&1; grep -A45 "renderFrameKdKernel<true, true, 0, false, 0, true, 0>" $O/r04_b_pmc_f$f/summary.txt | head -50; done
