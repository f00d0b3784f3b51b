#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then HBM-traffic PMC passes (separate runs, each
# under its own timeout).   usage: tools/prof_bench.sh <tag> [bench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/trace -- python3 bench.py --no-cpu-baseline --no-extras "$@" > gpurun_out/$tag/bench_trace.log 2>&1 || exit 1
timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$tag/fetch -- python3 bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 --repeats 1 "$@" > gpurun_out/$tag/bench_fetch.log 2>&1 || echo "FETCH_SIZE pass failed"
timeout -k 5 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$tag/write -- python3 bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 --repeats 1 "$@" > gpurun_out/$tag/bench_write.log 2>&1 || echo "WRITE_SIZE pass failed"
timeout -k 5 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/$tag/tcc -- python3 bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 --repeats 1 "$@" > gpurun_out/$tag/bench_tcc.log 2>&1 || echo "TCC pass failed"
timeout -k 5 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d gpurun_out/$tag/sq -- python3 bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 --repeats 1 "$@" > gpurun_out/$tag/bench_sq.log 2>&1 || echo "SQ pass failed"
timeout -k 5 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$tag/ta -- python3 bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 --repeats 1 "$@" > gpurun_out/$tag/bench_ta.log 2>&1 || echo "TA pass failed"
tail -1 gpurun_out/$tag/bench_trace.log | cut -c1-300
python3 tools/prof_report.py $tag
