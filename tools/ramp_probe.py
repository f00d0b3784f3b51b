"""How the search time settles after the device has been idle: a fresh handle per round (its construction and the filter
analysis leave the GPU idle for 0.1-0.3 s), then groups of five searches timed with HIP events, back to back.
usage: python tools/ramp_probe.py [protocol=CC11xx] [groups=8] [handles=3] [idle_s=0]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from pycusdr_amd.hostcpu import quiet_blas  # noqa: E402
quiet_blas()          # numpy's BLAS workers must not spend the container's CPU quota: a throttled host starves the device
from pycusdr_amd import config as cfg, signals as sg              # noqa: E402
from pycusdr_amd.demodulator.demodulator_base import doppler_bin_table   # noqa: E402
from pycusdr_amd.mfbank import MFBank                              # noqa: E402
from pycusdr_amd.protocol import loadProtocol                      # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'CC11xx'
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 8
handles = int(sys.argv[3]) if len(sys.argv) > 3 else 3
idle = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
log2N, D = 20, 256
N = 1 << log2N
if name == 'CC11xx':
    conf, sps, ms = cfg.cc11xx_config(blockSize=log2N, doppCarrierSteps=D, samplesPerSym=128), 128, 3
else:
    conf, sps, ms = cfg.bench_config(name, blockSize=log2N, doppCarrierSteps=D), 16, (5 if name == 'bench_BPSK' else 3)
_, _, shifts, _ = doppler_bin_table(conf['Radios']['Rx']['UHF-H'], conf['Radios']['rangeRateMax'], N)
M, masks = loadProtocol(name)(conf=conf).get_filter(N, sps, ms)
x = sg.s1_stream(1, N, 1 << 10, 'GMSK', snr_db=10.0, seed=1)[:N]
for it in range(handles):
    bank = MFBank(log2N, D, M)
    bank.set_filters(masks)
    bank.set_shifts(shifts)
    bank.upload(x)
    bank.find_carrier()
    for rnd in range(2 if idle else 1):
        ts = []
        for _ in range(groups):
            bank.timer_start()
            for _ in range(5):
                bank.search_async()
            ts.append(bank.timer_stop() / 5)
        print(f'{name} handle {it}{" after %.1f s idle" % idle if rnd else ""}: ' + ' '.join(f'{t:.3f}' for t in ts) + ' ms per search, groups of 5', flush=True)
        if idle:
            time.sleep(idle)
    bank.close()
