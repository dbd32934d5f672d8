#!/bin/bash
# round 5: large sweep of the seeded random families on fresh seeds, part 2
set -o pipefail
O=gpurun_out
cd tests
run() { name=$1; shift; timeout -k 10 "$TMO" python "$@" --keep-going > ../$O/r05_j_$name.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then echo "$name killed (rc $rc)"; exit $rc; fi; echo "$name: $(tail -1 ../$O/r05_j_$name.log)"; grep -m3 FAIL ../$O/r05_j_$name.log; }
TMO=400
run grids gpu_fuzz.py 6000 6999 --grids
run holes gpu_fuzz.py 6000 6599 --holes
run many gpu_fuzz.py 6000 6199 --many
run domains gpu_fuzz.py 6000 6299 --domains
run deep gpu_fuzz.py 6000 6039 --deep
run sched gpu_fuzz_sched.py 6000 6499
run state gpu_fuzz_state.py 6000 6299
run tracer gpu_fuzz_tracer.py 6000 6199
echo done
