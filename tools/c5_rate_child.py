"""One demodulator instance of BASELINE config C5 as a process of its own (the reference gives every radio its own process and
device context, pyCuSDR.py:245-251, demodulator_base.py:177-181): the search step of bench.py (forward FFT, Doppler search over D bins,
pick, 8-byte read-back) on blocks resident in HBM, for a set time from a set moment.  Started by bench.py's C5 leg (and
tools/c5_concurrent.py), alone and beside a sibling on the same device.
usage: c5_rate_child.py <CC11xx|bench_BPSK> <D> <seconds> [part parts]   -- prints 'ready', waits for 'go <epoch>' on stdin, prints one JSON
line; part / parts: the instance's share of the compute units (mfb_set_cu_share)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pycusdr_amd.hostcpu import quiet_blas          # noqa: E402
quiet_blas()
import torch                                          # noqa: E402
from pycusdr_amd import config as cfg               # noqa: E402
from pycusdr_amd.demodulator.demodulator_base import doppler_bin_table   # noqa: E402
from pycusdr_amd.mfbank import MFBank               # noqa: E402
from pycusdr_amd.protocol import loadProtocol       # noqa: E402

name, D, seconds = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
log2N, ov, NB = 20, 1 << 10, 4
N = 1 << log2N
if name == 'CC11xx':
    conf, sps, ms = cfg.cc11xx_config(blockSize=log2N, doppCarrierSteps=D), 128, 3
else:
    conf, sps, ms = cfg.bench_config(name, blockSize=log2N, doppCarrierSteps=D), 16, 5
_, _, shifts, _ = doppler_bin_table(conf['Radios']['Rx']['UHF-H'], conf['Radios']['rangeRateMax'], N)
M, masks = loadProtocol(name)(conf=conf).get_filter(N, sps, ms)
bank = MFBank(log2N, D, M)
if len(sys.argv) > 5:
    bank.set_cu_share(int(sys.argv[4]), int(sys.argv[5]))
bank.set_filters(masks)
bank.set_shifts(shifts)
rs = np.random.RandomState(5)
x = (rs.standard_normal((NB, N)) + 1j * rs.standard_normal((NB, N))).astype(np.complex64)        # S2: white noise, resident
dev = torch.from_numpy(x.view(np.float32).reshape(NB, 2 * N)).to('cuda:0')
torch.cuda.synchronize()
esz = 8 * N


def step(i):
    bank.upload_device(dev.data_ptr() + (i % NB) * esz)
    return bank.find_carrier()


step(0)
print('ready', flush=True)
go = sys.stdin.readline().split()
t_go = float(go[1])
# settle: untimed steps from 150 ms before the moment (the clock ramp after the wait, profiles/r03_ramp.md)
while time.time() < t_go - 0.15:
    time.sleep(0.005)
i = 0
while time.time() < t_go:
    step(i)
    i += 1
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < seconds:
    step(i)
    i += 1
    n += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({'protocol': name, 'D': D, 'M': M, 'steps': n, 'seconds': round(dt, 4), 'ms_per_step': round(dt / n * 1e3, 4),
                  'msamples': round((N - ov) * n / dt / 1e6, 2), 'path': bank.get_search_path(), 'started': t_go}), flush=True)
bank.close()
