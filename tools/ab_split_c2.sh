#!/bin/bash
# the whole receive chain at C2 (one block per call: 2^20-sample blocks), one stream against two, interleaved
for rep in 1 2; do
  for split in 0 1; do
    echo "== MFB_BATCH_SPLIT=$split rep $rep"
    MFB_BATCH_SPLIT=$split timeout -k 10 300 python3 tools/chain_rate.py 20 ${1:-120} 256 GMSK 1 2>&1 | grep "blocks_per_call"
  done
done
