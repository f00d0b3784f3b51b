#!/bin/bash
# The batch's two streams against one (MFB_BATCH_SPLIT=0), interleaved on one box: tools/ab_split.sh [log2N] [packets] [bins] [B list]
l=${1:-15}; n=${2:-240}; d=${3:-64}; bs=${4:-32,auto}
for rep in 1 2; do
  for split in 0 1; do
    echo "== MFB_BATCH_SPLIT=$split rep $rep"
    MFB_BATCH_SPLIT=$split timeout -k 10 300 python3 tools/chain_rate.py $l $n $d GMSK $bs 2>&1 | grep "blocks_per_call"
  done
done
