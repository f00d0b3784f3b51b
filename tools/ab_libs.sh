#!/bin/bash
# A/B of library builds on ONE box (devices differ by up to 10 %): tools/ab_libs.sh <probe args...> -- lib1 lib2 ...
args=(); libs=(); seen=0
for a in "$@"; do if [ "$a" == "--" ]; then seen=1; elif [ $seen == 0 ]; then args+=("$a"); else libs+=("$a"); fi; done
for rep in 1 2; do
  for lib in "${libs[@]}"; do
    if [ "$lib" == "default" ]; then unset MFBANK_LIB; else export MFBANK_LIB=$GRAFT_REPO_ROOT/$lib; fi
    echo "== $lib (rep $rep)"
    timeout -k 10 200 python tools/seg_probe.py "${args[@]}" 2>&1 | grep "segment\|twopass"
  done
done
