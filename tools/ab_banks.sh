#!/bin/bash
# A/B of library builds on ONE box over the three banks of the bench line (GMSK 48 taps, CC11xx 384 taps, BPSK 80 taps):
#   tools/ab_banks.sh build_var/lib_a.so build_var/lib_b.so ...      (interleaved, two repetitions; run on the GPU box)
libs=("$@")
for rep in 1 2; do
  for lib in "${libs[@]}"; do
    export MFBANK_LIB=$GRAFT_REPO_ROOT/$lib
    for bank in "bench_GMSK 8" "CC11xx 11" "bench_BPSK 8"; do
      name=${bank% *}; l=${bank#* }
      echo -n "== $lib rep $rep $name: "
      timeout -k 10 200 python tools/seg_probe.py 20 256 $name $l 32 --no-twopass 2>&1 | grep "^segment" | sed 's/parseval.*//'
    done
  done
done
