#!/bin/bash
# round 4: whole-frame oracle parity of C2 / C3 / C4 at BASELINE sizes, the ragged split + deferred-AO case, bench line with live PMC
set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_configs.py tests/test_gpu_interleave.py -x -q -s --durations=12 > $O/r04_d_tests.log 2>&1; tail -30 $O/r04_d_tests.log
python bench.py > $O/r04_d_bench.json 2> $O/r04_d_bench.err || tail -5 $O/r04_d_bench.err
tail -c 3000 $O/r04_d_bench.json
