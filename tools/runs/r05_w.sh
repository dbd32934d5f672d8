#!/bin/bash
# round 5, late: the whole GPU suite, the smoke entry and the seeded random families on fresh seeds with the seven-wave rope march
set -o pipefail
O=gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/r05_w_gpu_suite.log 2>&1; rc=$?; tail -4 $O/r05_w_gpu_suite.log
[ $rc -ge 124 ] && exit $rc
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_w_smoke.log 2>&1; tail -2 $O/r05_w_smoke.log
cd tests
run() { name=$1; shift; timeout -k 10 "$TMO" python "$@" --keep-going > ../$O/r05_w_$name.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then echo "$name killed (rc $rc)"; exit $rc; fi; echo "$name: $(tail -1 ../$O/r05_w_$name.log)"; grep -m3 FAIL ../$O/r05_w_$name.log; }
TMO=200
run plain gpu_fuzz.py 11000 11799
run rich gpu_fuzz.py 11000 11149 --rich
run grids gpu_fuzz.py 11000 11199 --grids
run holes gpu_fuzz.py 11000 11099 --holes
run sched gpu_fuzz_sched.py 11000 11099
run state gpu_fuzz_state.py 11000 11059
run many gpu_fuzz.py 11000 11149 --many
echo done
