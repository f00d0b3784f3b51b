#!/bin/bash
# one-block path at C2: one stream against two, interleaved: tools/ab_block_overlap.sh [log2N] [bins] [protocol]
for rep in 1 2; do
  for o in 0 1; do
    echo -n "== overlap=$o rep $rep: "
    timeout -k 10 200 python3 tools/block_device_rate.py ${1:-20} ${2:-256} 200 $o ${3:-bench_GMSK} 2>&1 | tail -1
  done
done
