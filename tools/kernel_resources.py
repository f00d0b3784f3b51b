#!/usr/bin/env python3
"""Registers / spills / occupancy of every kernel from a hipcc -Rpass-analysis=kernel-resource-usage log.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage ... 2> build.log ; tools/kernel_resources.py build.log [name-filter]"""
import re
import subprocess
import sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for b in re.split(r'remark: Function Name: ', text)[1:]:
    mangled = b.split()[0]
    if flt and flt not in mangled:
        continue
    try:
        name = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', mangled], capture_output=True, text=True).stdout.strip()
    except OSError:
        name = mangled
    def g(k):
        m = re.search(re.escape(k) + r': (\d+)', b)
        return m.group(1) if m else '?'
    print(f'{name[:60]:60s} VGPR {g("VGPRs"):>4} AGPR {g("AGPRs"):>3} spill {g("VGPRs Spill"):>3} SGPR {g("TotalSGPRs"):>3} '
          f'scratch {g("ScratchSize [bytes/lane]"):>4} occ {g("Occupancy [waves/SIMD]")}')
