#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/stall
timeout -k 5 150 rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL --output-format csv -d gpurun_out/stall/lds -- python3 bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 --repeats 1 --protocol CC11xx > gpurun_out/stall/lds.log 2>&1 || echo "lds pass failed"
timeout -k 5 150 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/stall/vmem -- python3 bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 --repeats 1 --protocol CC11xx > gpurun_out/stall/vmem.log 2>&1 || echo "vmem pass failed"
python3 - <<'PY'
import csv, glob, collections
for d in ('lds','vmem'):
    fs = sorted(glob.glob(f'gpurun_out/stall/{d}/*/*counter_collection.csv'))
    if not fs: print(d, 'no file'); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        if 'k_seg<2048, 0, 26>' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items(): print(d, k, '%.4g' % (sum(v)/len(v)), len(v))
PY
