#!/bin/bash
# A/B of library builds on ONE box, the 256-point workloads: tools/ab_banks2.sh lib1 lib2 ...   (interleaved, two repetitions)
for rep in 1 2; do
  for lib in "$@"; do
    export MFBANK_LIB=$GRAFT_REPO_ROOT/$lib
    for w in "bench_GMSK 256" "bench_BPSK 256" "bench_GMSK 1024"; do
      name=${w% *}; d=${w#* }
      echo -n "== $lib rep $rep $name D=$d: "
      timeout -k 10 200 python tools/seg_probe.py 20 $d $name 8 32 --no-twopass 2>&1 | grep "^segment" | sed 's/parseval.*//'
    done
  done
done
