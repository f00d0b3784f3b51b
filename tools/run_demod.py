"""A few blocks of find_carrier + demodulate + find_centres at C2 (for rocprofv3 / timing of the demodulation leg)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from pycusdr_amd import config as cfg, signals as sg
from pycusdr_amd.mfbank import MFBank
from pycusdr_amd.protocol import loadProtocol
from pycusdr_amd.demodulator.demodulator_base import doppler_bin_table
log2N, D = 20, 256
N = 1 << log2N
conf = cfg.bench_config('bench_GMSK', blockSize=log2N, doppCarrierSteps=D)
_, _, shifts, _ = doppler_bin_table(conf['Radios']['Rx']['UHF-H'], 7500, N)
M, masks = loadProtocol('bench_GMSK')(conf=conf).get_filter(N, 16, 3)
x = sg.s1_stream(1, N, 1 << 10, 'GMSK', snr_db=10.0, seed=1)[:N]
bank = MFBank(log2N, D, M)
bank.set_filters(masks); bank.set_shifts(shifts)
k_off = int(N / (1.1 * 16)); k_len = int(N / (0.9 * 16)) - k_off
bank.upload(x); bank.find_carrier()
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(10):
        k, arg, _ = bank.demodulate(N // 4, k_off, k_len)
        spS = N / float(k); cOff = -float(arg) / np.pi * spS / 2
        if cOff < 0: cOff += spS - 1
        bank.find_centres(np.float32(spS), np.float32(cOff), 0, int(N / spS))
    print('demodulate + find_centres: %.1f us per block' % ((time.perf_counter() - t0) / 10 * 1e6), flush=True)
bank.close()
