#!/bin/bash
# round 4: the channel-interleaved march at one more wave per SIMD (form 1 needs fewer registers): three channels at 5, two at 6
O=gpurun_out
run() { v=$1; shift; for rep in 1 2; do EXA_HIP_LIB=$PWD/build/variants/libexa_hip_$v.so python bench.py --cpu-baseline off --pmc off --steps 10 "$@" > $O/r04_n_tmp.json 2> $O/r04_n_tmp.err || tail -3 $O/r04_n_tmp.err; python -c "import json; d=json.loads(open('$O/r04_n_tmp.json').read().strip().splitlines()[-1]); print('$v $* : %.3f ms' % d['roofline']['kernel_ms'])"; done; }
run base --fields 3
run il34w5 --fields 3
run base --fields 4
run il34w5 --fields 4
run base --config c3_gear
run il2w6 --config c3_gear
run base --config c3_gear --iso 0.5
run il2w6 --config c3_gear --iso 0.5
