#!/bin/bash
# round 3, GPU run N: seeded parity sweeps with the final kernels (fresh seed ranges), then the final bench line
set -o pipefail
O=gpurun_out
cd tests
run() { name=$1; shift; python "$@" --keep-going > ../$O/r03_n_$name.log 2>&1; echo "$name: $(tail -1 ../$O/r03_n_$name.log)"; }
run plain gpu_fuzz.py 20000 20599
run rich gpu_fuzz.py 20000 20599 --rich
run grids gpu_fuzz.py 20000 20499 --grids
run many gpu_fuzz.py 20000 20599 --many
run deep gpu_fuzz.py 20000 20149 --deep
run domains gpu_fuzz.py 20000 20299 --domains
run clip gpu_fuzz.py 20000 20199 --rich --clip
run sched gpu_fuzz_sched.py 20000 20399
run state gpu_fuzz_state.py 20000 20199
run tracer gpu_fuzz_tracer.py 20000 20199
cd ..
python bench.py > $O/r03_n_bench.json 2> $O/r03_n_bench.err; tail -c 400 $O/r03_n_bench.json; echo
echo done
