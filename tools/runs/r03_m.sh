#!/bin/bash
set -o pipefail
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r03_m_tests.log 2>&1; tail -3 $O/r03_m_tests.log
python tools/walk_probe.py > $O/r03_m_walk_probe_c4.json 2>/dev/null; grep -E "restarts|lane_node_steps|wave_node|nodes_per" $O/r03_m_walk_probe_c4.json
bash tools/ab_variants.sh run --pmc off; cp $O/variants/results.txt $O/r03_m_ab_stack8_c4.txt
bash tools/ab_variants.sh run --pmc off --camera closeup; cp $O/variants/results.txt $O/r03_m_ab_stack8_c4_closeup.txt
bash tools/ab_variants.sh run --pmc off --config c3_gear --iso 0.5; cp $O/variants/results.txt $O/r03_m_ab_stack8_c3iso.txt
echo done
