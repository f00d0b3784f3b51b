#!/bin/bash
# k_segf<2048> on the 384-tap CC11xx bank: against seg_body, groups of slots / bins, rectangle sizes; interleaved on one box
for rep in 1 2; do
for v in "0 0 4,1" "1 1 2,1" "1 1 4,1" "1 1 8,1" "1 0 4,1" "1 1 4,2"; do set -- $v
 echo -n "== FSM=$1 GROUP=$2 rect=$3 CC11xx D=256: "; MFB_SEG_FSM=$1 MFB_SEG_FSM_GROUP=$2 MFB_SEG_FSM_RECT=$3 timeout -k 10 200 python tools/seg_probe.py 20 256 CC11xx 11 32 --no-twopass 2>&1 | grep "^segment"
done; done
