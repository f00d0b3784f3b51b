#!/bin/bash
# compare kernel-variant builds (build_var/lib_*.so) interleaved in one call
for r in 1 2; do for n in "$@"; do echo "== variant $n (round $r)"; MFBANK_LIB=$GRAFT_REPO_ROOT/build_var/lib_$n.so timeout -k 10 200 python tools/probe.py 20 256 8 2>&1 | grep -E "chunk|parseval" | head -3; done; done
