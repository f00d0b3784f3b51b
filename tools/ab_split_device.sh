#!/bin/bash
# device side of a batch, one stream against two, interleaved: tools/ab_split_device.sh [log2N] [bins] [B]
for rep in 1 2; do
  for split in 0 1; do
    echo "== MFB_BATCH_SPLIT=$split rep $rep"
    MFB_BATCH_SPLIT=$split timeout -k 10 200 python3 tools/batch_device_rate.py ${1:-15} ${2:-64} ${3:-32} 200 1 2>&1 | tail -1
  done
done
