#!/bin/bash
# round 4: -mllvm -amdgpu-sched-strategy=max-ilp on all kernels: frame times of several configurations, both builds interleaved
O=gpurun_out
for cfg in "" "--config c3_gear" "--config c3_gear --iso 0.5" "--fields 3" "--camera closeup" "--scale 1.25" "--size 4096 --iso 0.5 --ao --spp 4 --steps 3 --warmup 1"; do
  for v in base maxilp base maxilp; do
    EXA_HIP_LIB=$PWD/build/variants/libexa_hip_$v.so python bench.py --cpu-baseline off --pmc off --steps 10 $cfg > $O/r04_t_tmp.json 2>/dev/null && python -c "import json; d=json.loads(open('$O/r04_t_tmp.json').read().strip().splitlines()[-1]); print('$v [$cfg]: %.3f ms' % d['ms_per_step'])"
  done
done
