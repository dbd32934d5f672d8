#!/bin/bash
# round 5, last: GPU suite + smoke, the default bench line and the lines of the configurations on the final build
set -o pipefail
O=gpurun_out
stop() { rc=$1; if [ "$rc" -ge 124 ]; then echo "step killed (rc $rc): stopping"; exit "$rc"; fi; }
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/r05_zz_gpu_suite.log 2>&1; rc=$?; tail -3 $O/r05_zz_gpu_suite.log; stop $rc; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_zz_smoke.log 2>&1; tail -c 300 $O/r05_zz_smoke.log; echo
( time timeout -k 10 600 python bench.py > $O/r05_z_bench.json 2> $O/r05_z_bench.err ) 2> $O/r05_zz_bench.time; stop $?; grep real $O/r05_zz_bench.time
python -c "
import json; d=json.loads(open('gpurun_out/r05_z_bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print('bench: %.3f frames/s, %.3f ms per frame, one at a time %.3f, kernel %.3f (form 0 %.3f), valu %.3f, hbm %.3f' % (d['value'], d['ms_per_step'], d['latency_ms'], r['kernel_ms'], r['kernel_ms_basis_form0'], r['frac'], r['hbm_measured']['frac']))"
echo done
