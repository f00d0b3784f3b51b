#!/bin/bash
# S1 against S2 (and all-zero samples) on one box: interleaved timing, cycles per launch from the counters, watts and sclk beside long runs.
# usage (GPU box): bash tools/s1_vs_s2.sh [tag]      -> gpurun_out/<tag>/   (python3 tools/s1_vs_s2_report.py <tag> writes the summary)
tag=${1:-r06_s1s2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
mkdir -p $out
echo "== interleaved timing" | tee $out/progress.txt
timeout -k 5 240 python3 tools/s1_vs_s2.py ab bench_GMSK 256 3 40 > $out/ab_gmsk.txt 2>&1 || exit 1
tail -9 $out/ab_gmsk.txt
timeout -k 5 240 python3 tools/s1_vs_s2.py ab CC11xx 256 3 20 S1,S2,Z,S2q > $out/ab_cc11xx.txt 2>&1 || exit 1
tail -5 $out/ab_cc11xx.txt
for s in S1 S2 Z; do
  echo "== counters $s" | tee -a $out/progress.txt
  timeout -k 5 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc_$s -- python3 tools/s1_vs_s2.py one $s bench_GMSK 256 20 > $out/pmc_$s.log 2>&1 || echo "pmc $s failed"
  timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$s -- python3 tools/s1_vs_s2.py one $s bench_GMSK 256 200 > $out/trace_$s.log 2>&1 || echo "trace $s failed"
done
for s in S1 S2 Z; do
  echo "== power $s" | tee -a $out/progress.txt
  python3 tools/s1_vs_s2.py one $s bench_GMSK 256 6000 > $out/power_run_$s.log 2>&1 &
  pid=$!
  sleep 7
  for i in $(seq 1 10); do
    /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (junction|hotspot)" | tr '\n' ';'
    echo
    sleep 0.4
  done > $out/power_$s.txt
  wait $pid
  cat $out/power_run_$s.log | tail -1
  head -3 $out/power_$s.txt | cut -c1-300
done
echo "== done" | tee -a $out/progress.txt
