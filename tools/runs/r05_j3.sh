#!/bin/bash
# round 5: large sweep of the seeded random families on fresh seeds, part 3
set -o pipefail
O=gpurun_out
cd tests
run() { name=$1; shift; timeout -k 10 "$TMO" python "$@" --keep-going > ../$O/r05_j3_$name.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then echo "$name killed (rc $rc)"; exit $rc; fi; echo "$name: $(tail -1 ../$O/r05_j3_$name.log)"; grep -m3 FAIL ../$O/r05_j3_$name.log; }
TMO=800
run plain gpu_fuzz.py 8000 10999
TMO=300
run deep gpu_fuzz.py 6040 6139 --deep
run domains gpu_fuzz.py 6300 6799 --domains
run many gpu_fuzz.py 6200 6499 --many
echo done
