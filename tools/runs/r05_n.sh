#!/bin/bash
# round 5: the whole GPU suite and the smoke entry on the final state
set -o pipefail
O=gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/r05_n_gpu_suite.log 2>&1; rc=$?; tail -4 $O/r05_n_gpu_suite.log
[ $rc -ge 124 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_n_smoke.log 2>&1; tail -2 $O/r05_n_smoke.log
echo done
