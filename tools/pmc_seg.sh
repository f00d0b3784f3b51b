#!/bin/bash
# usage: tools/pmc_seg.sh <tag> <args to run_seg.py...>   (run on the GPU box)
# kernel trace, then SQ / LDS / TCC / HBM-traffic counters in separate passes.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/trace -- python3 tools/run_seg.py "$@" > gpurun_out/$tag/trace.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d gpurun_out/$tag/pmc1 -- python3 tools/run_seg.py "$@" > gpurun_out/$tag/pmc1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/$tag/pmc2 -- python3 tools/run_seg.py "$@" > gpurun_out/$tag/pmc2.log 2>&1 || exit 1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d gpurun_out/$tag/pmc3 -- python3 tools/run_seg.py "$@" > gpurun_out/$tag/pmc3.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$tag/pmc4 -- python3 tools/run_seg.py "$@" > gpurun_out/$tag/pmc4.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$tag/pmc5 -- python3 tools/run_seg.py "$@" > gpurun_out/$tag/pmc5.log 2>&1 || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d gpurun_out/$tag/pmc6 -- python3 tools/run_seg.py "$@" > gpurun_out/$tag/pmc6.log 2>&1
python3 tools/pmc_report.py $tag k_seg
