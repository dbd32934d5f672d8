#!/bin/bash
# round 5: the seeded random families on fresh seeds with the rope walk in every family (each seed: oracle once, three ways of
# finding the segments on the GPU, counters, shipped vs counting kernel, the defaults)
set -o pipefail
O=gpurun_out
cd tests
run() { name=$1; shift; timeout -k 10 "$TMO" python "$@" --keep-going > ../$O/r05_i_$name.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then echo "$name killed (rc $rc)"; exit $rc; fi; echo "$name: $(tail -1 ../$O/r05_i_$name.log)"; grep -m3 FAIL ../$O/r05_i_$name.log; }
TMO=900
run plain gpu_fuzz.py 5000 5299
run rich gpu_fuzz.py 5000 5199 --rich
run grids gpu_fuzz.py 5000 5199 --grids
run holes gpu_fuzz.py 5000 5149 --holes
run sched gpu_fuzz_sched.py 5000 5149
run state gpu_fuzz_state.py 5000 5099
echo done
