#!/bin/bash
# Kernel trace of the batched receive chain at a reference block size (device time per block, launches per batch).
#   usage (on the GPU box): tools/chain_kernels.sh <tag> [log2N] [bins] [blocks per call] [GMSK|FSK|GFSK|BPSK|CC11xx] [packets]
tag=$1; n=${2:-15}; d=${3:-64}; b=${4:-16}; mod=${5:-GMSK}; pk=${6:-120}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/trace -- python3 tools/chain_rate.py $n $pk $d $mod $b > gpurun_out/$tag/chain_trace.log 2>&1 || exit 1
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
for f in sorted(glob.glob(f'gpurun_out/{tag}/trace/*/*kernel_stats.csv')):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(f'total kernel time {tot / 1e6:.2f} ms')
    print('kernel,calls,avg_us,total_ms,pct')
    for r in rows:
        if float(r['Percentage']) > 0.3:
            print(f"{r['Name'][:64]},{r['Calls']},{float(r['AverageNs']) / 1e3:.1f},{float(r['TotalDurationNs']) / 1e6:.2f},{r['Percentage']}")
PY
