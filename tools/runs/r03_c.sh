#!/bin/bash
# round 3, GPU run C: brick_order tests, Morton order A/B at scale 1.0 / 1.25, PMC of the interleaved configurations
set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_interleave.py -x -q > $O/r03_c_tests.log 2>&1; tail -3 $O/r03_c_tests.log
b() { name=$1; shift; python bench.py --steps 10 --cpu-baseline off --pmc on "$@" > $O/r03_c_$name.json 2> $O/r03_c_$name.err || tail -3 $O/r03_c_$name.err; }
b c4_s100_morton --option brick_order=1
b c4_s125_order0 --scale 1.25
b c4_s125_morton --scale 1.25 --option brick_order=1
b f3_il1 --fields 3
b c3_il1 --config c3_gear
b c3iso_il1 --config c3_gear --iso 0.5
echo done
