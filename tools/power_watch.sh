#!/bin/bash
# Sample power / clocks with rocm-smi while the bench's timed loop runs.  usage: tools/power_watch.sh [bench args...]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 bench.py --no-cpu-baseline --no-extras --steps 9000 --warmup 20 "$@" > gpurun_out/power_bench.log 2>&1 &
pid=$!
sleep 9
for i in $(seq 1 16); do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (edge|junction|hotspot)" | tr '\n' ';'
  echo
  sleep 0.5
done > gpurun_out/power_samples.txt
wait $pid
tail -1 gpurun_out/power_bench.log | cut -c1-200
cat gpurun_out/power_samples.txt | cut -c1-400
