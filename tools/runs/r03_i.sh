#!/bin/bash
# round 3, GPU run I: the round's reference measurements with the final kernels
set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r03_i_tests.log 2>&1; tail -3 $O/r03_i_tests.log
python bench.py > $O/r03_i_bench.json 2> $O/r03_i_bench.err; tail -c 600 $O/r03_i_bench.json; echo
bash tools/config_timeline.sh $O/r03_i_tl_c4 --steps 10 --pmc off
bash tools/pmc_run.sh $O/r03_i_pmc_c4 > $O/r03_i_pmc_c4.txt 2>&1; tail -40 $O/r03_i_pmc_c4.txt
bash tools/pmc_run.sh $O/r03_i_pmc_c3iso --config c3_gear --iso 0.5 > $O/r03_i_pmc_c3iso.txt 2>&1; tail -5 $O/r03_i_pmc_c3iso.txt
python tests/gpu_shard_scaling.py 1.0 4 > $O/r03_i_shard_scaling.txt 2>&1; cat $O/r03_i_shard_scaling.txt
bash tools/rehearse_ranks.sh $O/r03_i_rehearse4 4 0.5 > $O/r03_i_rehearse4.txt 2>&1; cat $O/r03_i_rehearse4.txt
echo done
