"""How evenly does the search grid drain?  Needs a variant build with -DMFB_SEG_TRACE (start / end time and XCC of every
workgroup of the branch-free search kernel):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -pthread -DMFB_SEG_TRACE -o build_var/libmfbank_trace.so pycusdr_amd/csrc/mfbank.hip
    MFBANK_LIB=build_var/libmfbank_trace.so python tools/xcd_trace.py [wg_per_cu] [D]
Prints, per XCC, the workgroups it ran, when its last one ended and its summed busy time, relative to the kernel."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, '.')
from pycusdr_amd import config as cfg, signals as sg, _lib
from pycusdr_amd.mfbank import MFBank
from pycusdr_amd.protocol import loadProtocol
from pycusdr_amd.demodulator.demodulator_base import doppler_bin_table

wpc = int(sys.argv[1]) if len(sys.argv) > 1 else 0
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
log2N = 20
N = 1 << log2N
conf = cfg.bench_config('bench_GMSK', blockSize=log2N, doppCarrierSteps=D)
_, _, shifts, _ = doppler_bin_table(conf['Radios']['Rx']['UHF-H'], 7500, N)
M, masks = loadProtocol('bench_GMSK')(conf=conf).get_filter(N, 16, 3)
x = sg.s1_stream(1, N, 1 << 10, 'GMSK', snr_db=10.0, seed=1)[:N]
bank = MFBank(log2N, D, M)
bank.set_filters(masks)
bank.set_shifts(shifts)
bank.set_search_path('segment', 8, wpc)
bank.upload(x)
for _ in range(5):
    bank.find_carrier()
lib = _lib.load()
lib.mfb_debug_read_trace.argtypes = [C.c_void_p, C.c_int]
nb = 65536
buf = np.zeros((nb, 3), dtype=np.uint64)
assert lib.mfb_debug_read_trace(buf.ctypes.data, nb) == 0
used = buf[:, 1] > 0
t0, t1, xcc = buf[used, 0].astype(np.int64), buf[used, 1].astype(np.int64), buf[used, 2].astype(np.int64) & 0xF
start, end = t0.min(), t1.max()
span = float(end - start)
print(f'{used.sum()} workgroups, kernel span {span / 100:.1f} us (100 MHz clock), blockIdx % 8 == XCC for {np.mean((np.nonzero(used)[0] % 8) == xcc) * 100:.1f} % of them')
idx = np.nonzero(used)[0]
for k in range(8):
    sel = xcc == k
    if not sel.any():
        continue
    print(f'XCC {k} runs segment group(s) {sorted(set((idx[sel] % 8).tolist()))}', end=';  ')
    print(f'XCC {k}: {sel.sum():5d} workgroups, first start {(t0[sel].min() - start) / span * 100:5.1f} %, last end {(t1[sel].max() - start) / span * 100:6.2f} % of the span, '
          f'mean workgroup time {np.mean(t1[sel] - t0[sel]) / 100:7.1f} us, busy sum {np.sum(t1[sel] - t0[sel]) / 100 / 1e3:8.2f} ms')
bank.close()
