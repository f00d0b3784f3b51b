"""Time the opt-in spectral-energy search against the default search at full size.  usage: energy_probe.py [D] [protocol] [sum_all]"""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
from pycusdr_amd import config as cfg
from pycusdr_amd.protocol import loadProtocol
from pycusdr_amd.mfbank import MFBank

D = int(sys.argv[1]) if len(sys.argv) > 1 else 256
name = sys.argv[2] if len(sys.argv) > 2 else 'bench_GMSK'
sum_all = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
log2N = 20
N = 1 << log2N
conf = cfg.bench_config(name, blockSize=log2N, doppCarrierSteps=D)
proto = loadProtocol(name)(conf=conf)
sps = 128 if name.startswith('CC11xx') else 16
msz = 5 if name == 'bench_BPSK' else 3
M, masks = proto.get_filter(N, sps, msz)
rs = np.random.RandomState(0)
x = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
shifts = np.round(np.linspace(0.25 - 0.07, 0.25 + 0.07, D) * N).astype(np.int32)
bank = MFBank(log2N, D, M, sum_all_masks=sum_all)
bank.set_filters(masks)
bank.set_shifts(shifts)
bank.upload(x)
out = {}
for mode in ('transforms', 'energy'):
    bank.set_search_mode(mode)
    for _ in range(3):
        bank.find_carrier()
    t0 = time.perf_counter()
    n = 20 if mode == 'transforms' else 200
    for _ in range(n):
        res = bank.find_carrier()
    dt = (time.perf_counter() - t0) / n
    t1 = time.perf_counter()
    for _ in range(n):
        bank.upload(x)
        res = bank.find_carrier()
    dt2 = (time.perf_counter() - t1) / n
    out[mode] = (dt, dt2, res, bank.get_scores())
    print(f'{mode:10s}: search+pick {dt * 1e3:.3f} ms   upload+fft+search+pick {dt2 * 1e3:.3f} ms ({(N - 1024) / dt2 / 1e6:.0f} Msamples/s)  idx {res[0]:.4f}')
a, b = out['transforms'][3], out['energy'][3]
print('max rel diff energy vs transforms:', float(np.abs(a - b).max() / a.max()))
bank.close()
