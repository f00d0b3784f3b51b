#!/bin/bash
# The search with the shift on the filter side (k_segf) against seg_body, and its rectangle sizes, interleaved on one box:
#   tools/ab_fsm.sh [protocol] [D]
name=${1:-bench_GMSK}; D=${2:-256}
for rep in 1 2; do
  for v in "0 8,1" "1 4,1" "1 8,1" "1 16,1" "1 8,2" "1 32,1"; do
    set -- $v
    echo -n "== FSM=$1 rect=$2 rep $rep $name D=$D: "
    MFB_SEG_FSM=$1 MFB_SEG_FSM_RECT=$2 timeout -k 10 200 python tools/seg_probe.py 20 $D $name 8 32 --no-twopass 2>&1 | grep "^segment"
  done
done
