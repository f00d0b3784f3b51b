#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then HBM-traffic PMC passes (separate runs).
# usage: tools/prof_bench.sh <tag> [bench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/trace -- python3 bench.py --no-cpu-baseline --no-demod-leg "$@" > gpurun_out/$tag/bench_trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$tag/fetch -- python3 bench.py --no-cpu-baseline --no-demod-leg --steps 4 --warmup 1 > gpurun_out/$tag/bench_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$tag/write -- python3 bench.py --no-cpu-baseline --no-demod-leg --steps 4 --warmup 1 > gpurun_out/$tag/bench_write.log 2>&1 || exit 1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/$tag/tcc -- python3 bench.py --no-cpu-baseline --no-demod-leg --steps 4 --warmup 1 > gpurun_out/$tag/bench_tcc.log 2>&1 || exit 1
tail -1 gpurun_out/$tag/bench_trace.log
