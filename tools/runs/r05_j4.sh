#!/bin/bash
# round 5: large sweep of the seeded random families on fresh seeds, part 4
set -o pipefail
O=gpurun_out
cd tests
run() { name=$1; shift; timeout -k 10 "$TMO" python "$@" --keep-going > ../$O/r05_j4_$name.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then echo "$name killed (rc $rc)"; exit $rc; fi; echo "$name: $(tail -1 ../$O/r05_j4_$name.log)"; grep -m3 FAIL ../$O/r05_j4_$name.log; }
TMO=500
run rich gpu_fuzz.py 7000 8499 --rich
run grids gpu_fuzz.py 7000 8499 --grids
TMO=300
run holes gpu_fuzz.py 7000 7799 --holes
run sched gpu_fuzz_sched.py 7000 7799
run state gpu_fuzz_state.py 7000 7399
echo done
