#!/bin/bash
# round 3, GPU run D: full suite after the pre-pass changes, walk probe, kernel timelines of C3 / C5, A/B of wave counts
set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r03_d_tests.log 2>&1; tail -3 $O/r03_d_tests.log
python tools/walk_probe.py > $O/r03_d_walk_probe_c4.json 2> $O/r03_d_walk_probe_c4.err; cat $O/r03_d_walk_probe_c4.json
python tools/walk_probe.py --camera closeup > $O/r03_d_walk_probe_c4_closeup.json 2>> $O/r03_d_walk_probe_c4.err
bash tools/config_timeline.sh $O/r03_d_tl_c3iso --config c3_gear --iso 0.5 --steps 10 --pmc off
bash tools/config_timeline.sh $O/r03_d_tl_c5 --size 4096 --iso 0.5 --ao --spp 16 --steps 2 --warmup 1 --pmc off
bash tools/ab_variants.sh run --config c3_gear --iso 0.5 --pmc off; cp $O/variants/results.txt $O/r03_d_ab_c3iso.txt
rm -f build/variants/libexa_hip_piw*.so
bash tools/ab_variants.sh run --fields 3 --pmc off; cp $O/variants/results.txt $O/r03_d_ab_f3.txt
echo done
