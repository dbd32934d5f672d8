#!/bin/bash
# round 4: the seeded random families on fresh seeds with the shipped default (basis_form 1 on both sides), and a slice in form 0
O=gpurun_out
cd tests
run() { name=$1; shift; python gpu_fuzz.py "$@" --keep-going > ../$O/r04_p_$name.log 2>&1; echo "$name: $(tail -1 ../$O/r04_p_$name.log)"; }
run plain 60000 60599
run rich 60000 60399 --rich
run grids 60000 60399 --grids
run many 60000 60199 --many
run deep 60000 60099 --deep
run domains 60000 60199 --domains
run clip 60000 60199 --rich --clip
run holes 60000 60399 --holes
EXA_TEST_BASIS_FORM=0 python gpu_fuzz.py 61000 61299 --keep-going > ../$O/r04_p_plain_form0.log 2>&1; echo "plain, form 0: $(tail -1 ../$O/r04_p_plain_form0.log)"
EXA_TEST_BASIS_FORM=0 python gpu_fuzz.py 61000 61199 --rich --keep-going > ../$O/r04_p_rich_form0.log 2>&1; echo "rich, form 0: $(tail -1 ../$O/r04_p_rich_form0.log)"
python gpu_fuzz_sched.py 60000 60299 > ../$O/r04_p_sched.log 2>&1; echo "sched: $(tail -1 ../$O/r04_p_sched.log)"
python gpu_fuzz_state.py 60000 60149 > ../$O/r04_p_state.log 2>&1; echo "state: $(tail -1 ../$O/r04_p_state.log)"
python gpu_fuzz_tracer.py 60000 60199 > ../$O/r04_p_tracer.log 2>&1; echo "tracer: $(tail -1 ../$O/r04_p_tracer.log)"
