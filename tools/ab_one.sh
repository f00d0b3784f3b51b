#!/bin/bash
# A/B of library builds on ONE box for ONE bank: tools/ab_one.sh <protocol> <log2L> lib1 lib2 ...   (interleaved, three repetitions)
name=$1; l=$2; shift 2
for rep in 1 2 3; do
  for lib in "$@"; do
    export MFBANK_LIB=$GRAFT_REPO_ROOT/$lib
    echo -n "== $lib rep $rep $name: "
    timeout -k 10 200 python tools/seg_probe.py 20 256 $name $l 32 --no-twopass 2>&1 | grep "^segment" | sed 's/parseval.*//'
  done
done
