"""One C2-style block through the search at one segment length / decomposition (for rocprofv3).
usage: python tools/run_seg.py <log2L|0=twopass> [wpc] [fpp] [D] [protocol] [reps]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from pycusdr_amd import config as cfg
from pycusdr_amd.mfbank import MFBank
from pycusdr_amd.protocol import loadProtocol
from pycusdr_amd.demodulator.demodulator_base import doppler_bin_table

l = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wpc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
fpp = int(sys.argv[3]) if len(sys.argv) > 3 else 0
D = int(sys.argv[4]) if len(sys.argv) > 4 else 256
name = sys.argv[5] if len(sys.argv) > 5 else 'bench_GMSK'
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
log2N = 20
N = 1 << log2N
if name == 'CC11xx':
    conf, sps, ms = cfg.cc11xx_config(blockSize=log2N, doppCarrierSteps=D), 128, 3
else:
    conf, sps, ms = cfg.bench_config(name, blockSize=log2N, doppCarrierSteps=D), 16, (5 if name == 'bench_BPSK' else 3)
_, _, shifts, _ = doppler_bin_table(conf['Radios']['Rx']['UHF-H'], conf['Radios']['rangeRateMax'], N)
M, masks = loadProtocol(name)(conf=conf).get_filter(N, sps, ms)
rs = np.random.RandomState(0)
x = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
bank = MFBank(log2N, D, M)
bank.set_filters(masks)
bank.set_shifts(shifts)
bank.upload(x)
if l:
    bank.set_search_path('segment', l, wpc, fpp)
else:
    bank.set_search_path('twopass')
print(bank.get_search_path())
for _ in range(reps):
    print(bank.find_carrier())
bank.close()
