#!/bin/bash
# The round's evidence in one call on ONE GPU box: the un-profiled bench line, then the kernel-trace and PMC passes of its three
# banks and of the two-pass path, then the batched receive chain's kernel trace.  usage: tools/round_profiles.sh r5
# (afterwards, here: python tools/pmc_summary.py r5 r05)
tag=${1:-r5}
mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { echo "bench failed"; tail -5 gpurun_out/${tag}_bench.err; exit 1; }
echo "bench done"
tools/prof_bench.sh $tag > gpurun_out/${tag}_report.txt 2>&1 && echo "prof C2 done"
tools/prof_bench.sh ${tag}_cc --protocol CC11xx > gpurun_out/${tag}_cc_report.txt 2>&1 && echo "prof CC11xx done"
tools/prof_bench.sh ${tag}_bpsk --protocol bench_BPSK > gpurun_out/${tag}_bpsk_report.txt 2>&1 && echo "prof BPSK done"
tools/prof_bench.sh ${tag}_twopass --path twopass --steps 12 > gpurun_out/${tag}_twopass_report.txt 2>&1 && echo "prof twopass done"
tools/chain_kernels.sh ${tag}_chain15 15 64 16 > gpurun_out/${tag}_chain15_kernels.txt 2>&1 && echo "chain 2^15 done"
tools/chain_kernels.sh ${tag}_chain17 17 64 8 > gpurun_out/${tag}_chain17_kernels.txt 2>&1 && echo "chain 2^17 done"
