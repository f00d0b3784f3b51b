#!/bin/bash
# the batched receive loop with one, two, three host copy threads (MFB_HOSTCOPY_THREADS), interleaved on one box
for rep in 1 2; do for t in 1 2 3; do
  echo "== MFB_HOSTCOPY_THREADS=$t rep $rep"
  MFB_HOSTCOPY_THREADS=$t timeout -k 10 300 python3 tools/chain_rate.py ${1:-15} ${2:-240} 64 GMSK ${3:-32} 2>&1 | grep blocks_per_call
done; done
