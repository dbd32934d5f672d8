#!/bin/bash
# round 4: is the vector L1 path (TA / TCP) what the march waits for?  busy and stall counters of the bench frame
# (two counters per hardware block per pass; the first refused pass ends the script)
O=gpurun_out
K="form1::renderFrameKdKernel<true, true, 0, false, 0, true, 0>"
run() { bash tools/pmc_extra.sh $O/r04_e_$1 "$2" --pmc off > $O/r04_e_$1.log 2>&1 || { tail -3 $O/r04_e_$1.log; return 1; }; grep -A12 "$K" $O/r04_e_$1/summary.txt | head -12; }
run ta1 "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" &&
run ta2 "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" &&
run ta3 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" &&
run tcp1 "TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" &&
run tcp2 "TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_ACCESSES_sum" &&
run tcp3 "TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum" &&
run tcp4 "TCP_TAGRAM0_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" &&
run td1 "TD_TD_BUSY_sum TD_TC_STALL_sum"
