#!/bin/bash
# k_segf: groups of slots against groups of bins (MFB_SEG_FSM_GROUP=0/1) and rectangle sizes, three workloads, interleaved on one box
for rep in 1 2; do
for v in "0 8,1" "1 8,1" "1 16,1" "0 16,1"; do set -- $v
 echo -n "== GROUP=$1 rect=$2 GMSK D=256: "; MFB_SEG_FSM_GROUP=$1 MFB_SEG_FSM_RECT=$2 timeout -k 10 200 python tools/seg_probe.py 20 256 bench_GMSK 8 32 --no-twopass 2>&1 | grep "^segment"
 echo -n "== GROUP=$1 rect=$2 BPSK: "; MFB_SEG_FSM_GROUP=$1 MFB_SEG_FSM_RECT=$2 timeout -k 10 200 python tools/seg_probe.py 20 256 bench_BPSK 8 32 --no-twopass 2>&1 | grep "^segment"
 echo -n "== GROUP=$1 rect=$2 GMSK D=1024: "; MFB_SEG_FSM_GROUP=$1 MFB_SEG_FSM_RECT=$2 timeout -k 10 200 python tools/seg_probe.py 20 1024 bench_GMSK 8 32 --no-twopass 2>&1 | grep "^segment"
 echo -n "== GROUP=$1 rect=$2 chain dev 2^15x64 B=32: "; MFB_SEG_FSM_GROUP=$1 MFB_SEG_FSM_RECT=$2 timeout -k 10 200 python3 tools/batch_device_rate.py 15 64 32 200 1 0 2>&1 | tail -1
done; done
