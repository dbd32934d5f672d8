#!/bin/bash
# round 4: scenes with empty cells (ALLOW_EMPTY_CELLS) on the GPU + a sweep of the new fuzz family; the whole GPU suite
set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_empty_cells.py -x -q > $O/r04_j_empty.log 2>&1; tail -4 $O/r04_j_empty.log
python tests/gpu_fuzz.py 100 299 --holes --keep-going > $O/r04_j_fuzz_holes.log 2>&1; tail -3 $O/r04_j_fuzz_holes.log
python tests/gpu_fuzz.py 100 199 --holes --rich --keep-going > $O/r04_j_fuzz_holes_rich.log 2>&1; tail -3 $O/r04_j_fuzz_holes_rich.log
python tests/gpu_fuzz.py 100 199 --holes --grids --keep-going > $O/r04_j_fuzz_holes_grids.log 2>&1; tail -3 $O/r04_j_fuzz_holes_grids.log
python -m pytest tests -q -m gpu -x > $O/r04_j_gpu_suite.log 2>&1; tail -4 $O/r04_j_gpu_suite.log
