#!/bin/bash
# usage: tools/pmc_ta.sh <tag> <args to run_seg.py...>  -- texture-addresser / L1 counters of the search kernels
# (two counters of a block per pass: the TA/TCP/TD blocks have few slots; every pass under its own timeout)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
i=0
for set in "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum" \
           "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL"; do
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --pmc $set --output-format csv -d gpurun_out/$tag/pmc$i -- python3 tools/run_seg.py "$@" > gpurun_out/$tag/pmc$i.log 2>&1 || echo "pass $i ($set) failed"
done
python3 tools/pmc_report.py $tag k_seg
