#!/bin/bash
# round 5, last: the seeded random families on fresh seeds with the final build
set -o pipefail
O=gpurun_out
cd tests
run() { name=$1; shift; timeout -k 10 "$TMO" python "$@" --keep-going > ../$O/r05_w2_$name.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then echo "$name killed (rc $rc)"; exit $rc; fi; echo "$name: $(tail -1 ../$O/r05_w2_$name.log)"; grep -m3 FAIL ../$O/r05_w2_$name.log; }
TMO=200
run plain gpu_fuzz.py 13000 13599
run rich gpu_fuzz.py 13000 13149 --rich
run grids gpu_fuzz.py 13000 13149 --grids
run holes gpu_fuzz.py 13000 13099 --holes
run state gpu_fuzz_state.py 13000 13059
run many gpu_fuzz.py 13000 13099 --many
run deep gpu_fuzz.py 13000 13039 --deep
echo done
