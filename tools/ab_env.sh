#!/bin/bash
mkdir -p gpurun_out
# A/B of an environment switch of the library (e.g. MFB_NO_SIDE, MFB_NO_GRAPH) on ONE box, whole bench lines interleaved:
#   tools/ab_env.sh MFB_NO_SIDE [reps]        (run on the GPU box; prints value / ms_per_step / chain figures per run)
var=$1; reps=${2:-2}
for rep in $(seq $reps); do
  for val in 1 0; do
    export $var=$val
    python bench.py --steps 20 --warmup 5 > gpurun_out/ab_env_$val.json 2> gpurun_out/ab_env_$val.err || { echo "bench failed ($var=$val)"; tail -5 gpurun_out/ab_env_$val.err; exit 1; }
    python - "$var=$val rep $rep" gpurun_out/ab_env_$val.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); c = d["config"]
print(sys.argv[1], "C2 %.4f ms" % d["ms_per_step"], "cc11xx %.4f" % c["cc11xx_ms_per_step"], "bpsk %.4f" % c["bpsk_ms_per_step"],
      "c3 %.4f" % c["c3_ms_per_step"], "one-call %.4f" % c["receive_block_one_call_ms"], "fc+demod %.4f" % c["find_carrier_plus_demodulate_ms"])
PY
  done
done
