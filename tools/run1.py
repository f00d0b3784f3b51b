"""Scratch: run find_carrier a few times at one tuning (for rocprofv3)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from pycusdr_amd.mfbank import MFBank
log2N, D, M = 20, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 8
tun = [int(a) for a in sys.argv[2:6]] if len(sys.argv) > 5 else [32, 8, 32, 8]
N = 1 << log2N
rs = np.random.RandomState(0)
x = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
masks = (rs.standard_normal((M, N)) + 1j * rs.standard_normal((M, N))).astype(np.complex64)
shifts = np.sort(rs.choice(N, D, replace=False)).astype(np.int32)
bank = MFBank(log2N, D, M)
bank.set_filters(masks); bank.set_shifts(shifts); bank.upload(x)
bank.set_tuning(*tun)
for _ in range(3):
    print(bank.find_carrier())
bank.close()
