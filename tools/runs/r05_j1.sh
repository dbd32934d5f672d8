#!/bin/bash
# round 5: large sweep of the seeded random families on fresh seeds, part 1 (see r05_i.sh)
set -o pipefail
O=gpurun_out
cd tests
run() { name=$1; shift; timeout -k 10 "$TMO" python "$@" --keep-going > ../$O/r05_j_$name.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then echo "$name killed (rc $rc)"; exit $rc; fi; echo "$name: $(tail -1 ../$O/r05_j_$name.log)"; grep -m3 FAIL ../$O/r05_j_$name.log; }
TMO=700
run plain gpu_fuzz.py 6000 7999
run rich gpu_fuzz.py 6000 6999 --rich
echo done
