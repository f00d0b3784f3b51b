#!/bin/bash
# Which device is this, and how fast are the three segment kernels on it at the settled clock?  One line per box for
# profiles/r03_ramp.md: GPU uuid, firmware versions, ms per search of the GMSK (256 points) and CC11xx (2048 / 4096 points) banks.
uuid=$(rocminfo 2>/dev/null | grep -i "Uuid:.*GPU" | head -1 | awk '{print $2}')
fw=$(rocm-smi --showfwinfo 2>/dev/null | grep -iE "MEC |MEC2|RLC |SMC|SDMA |PSP SOS|VBIOS|TA XGMI" | sed 's/GPU\[0\]\s*:\s*//' | tr -s ' \t' ' ' | tr '\n' ';')
vb=$(rocm-smi --showvbios 2>/dev/null | grep -i vbios | sed 's/GPU\[0\]\s*:\s*//' | tr -s ' \t' ' ' | head -1)
a=$(python tools/seg_probe.py 20 256 bench_GMSK 8 32 --no-twopass 2>&1 | grep "^segment" | awk '{print $5}')
b=$(python tools/seg_probe.py 20 256 CC11xx 11,12 32 --no-twopass 2>&1 | grep "^segment" | awk '{print $5}' | tr '\n' ' ')
echo "BOX $uuid | L256 $a | L2048 L4096 $b | $vb | $fw"
