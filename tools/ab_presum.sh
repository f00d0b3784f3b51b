#!/bin/bash
# A/B of the per-bin Parseval total (MFB_SEG_SUMQ) on ONE box: three banks, interleaved; full probe lines (parity figures included)
for spec in "bench_GMSK 8" "CC11xx 11" "bench_BPSK 8"; do
  set -- $spec
  for rep in 1 2 3; do
    for lib in build_var/lib_sumq.so pycusdr_amd/libmfbank.so; do
      export MFBANK_LIB=$GRAFT_REPO_ROOT/$lib
      echo -n "== $lib rep $rep $1: "
      timeout -k 10 200 python tools/seg_probe.py 20 256 $1 $2 32 --no-twopass 2>&1 | grep "^segment"
    done
  done
done
