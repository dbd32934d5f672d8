#!/bin/bash
# round 4: the seeded random families on fresh seeds with the shipped default (basis_form 1 on both sides), and a slice in form 0
O=gpurun_out
B=${SEED_BASE:-6}       # seeds ${B}0000 ..: SEED_BASE=7 for the second sweep (after form 1 was extended to the epilogue)
cd tests
run() { name=$1; shift; python gpu_fuzz.py "$@" --keep-going > ../$O/r04_p${B}_$name.log 2>&1; echo "$name: $(tail -1 ../$O/r04_p${B}_$name.log)"; }
run plain ${B}0000 ${B}0599
run rich ${B}0000 ${B}0399 --rich
run grids ${B}0000 ${B}0399 --grids
run many ${B}0000 ${B}0199 --many
run deep ${B}0000 ${B}0099 --deep
run domains ${B}0000 ${B}0199 --domains
run clip ${B}0000 ${B}0199 --rich --clip
run holes ${B}0000 ${B}0399 --holes
EXA_TEST_BASIS_FORM=0 python gpu_fuzz.py ${B}1000 ${B}1299 --keep-going > ../$O/r04_p${B}_plain_form0.log 2>&1; echo "plain, form 0: $(tail -1 ../$O/r04_p${B}_plain_form0.log)"
EXA_TEST_BASIS_FORM=0 python gpu_fuzz.py ${B}1000 ${B}1199 --rich --keep-going > ../$O/r04_p${B}_rich_form0.log 2>&1; echo "rich, form 0: $(tail -1 ../$O/r04_p${B}_rich_form0.log)"
python gpu_fuzz_sched.py ${B}0000 ${B}0299 > ../$O/r04_p${B}_sched.log 2>&1; echo "sched: $(tail -1 ../$O/r04_p${B}_sched.log)"
python gpu_fuzz_state.py ${B}0000 ${B}0149 > ../$O/r04_p${B}_state.log 2>&1; echo "state: $(tail -1 ../$O/r04_p${B}_state.log)"
python gpu_fuzz_tracer.py ${B}0000 ${B}0199 > ../$O/r04_p${B}_tracer.log 2>&1; echo "tracer: $(tail -1 ../$O/r04_p${B}_tracer.log)"
