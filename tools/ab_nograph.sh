#!/bin/bash
# the batched receive loop with and without the recorded graphs (MFB_NO_GRAPH), interleaved on one box
for rep in 1 2; do for g in 0 1; do
  echo "== MFB_NO_GRAPH=$g rep $rep"
  MFB_NO_GRAPH=$g timeout -k 10 300 python3 tools/chain_rate.py ${1:-15} ${2:-240} 64 GMSK ${3:-32} 2>&1 | grep blocks_per_call
done; done
