"""Throughput of the batched sync/preamble correlator: B=1024 bit streams."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from pycusdr_amd.mfbank import sync_correlate
rs = np.random.RandomState(2)
B, L, T = 1024, 65536 + 2048, 64
bits = rs.randint(0, 2, (B, L)).astype(np.uint8)
tmpl = (rs.randint(0, 2, T) * 2 - 1).astype(np.int8)
sync_correlate(bits[:2], tmpl)
t = time.perf_counter(); out = sync_correlate(bits, tmpl); dt = time.perf_counter() - t
t = time.perf_counter(); ref = [np.convolve(bits[b].astype(np.int64), tmpl.astype(np.int64)) for b in range(8)]; dc = (time.perf_counter() - t) / 8
assert all(np.array_equal(out[b], ref[b]) for b in range(8))
print(f'GPU (host in/out, int32 scores): {dt*1e3:.1f} ms for {B} streams -> {B/dt:.0f} streams/s; np.convolve: {dc*1e3:.2f} ms/stream -> {1/dc:.0f} streams/s/core')
