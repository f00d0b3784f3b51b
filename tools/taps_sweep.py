"""Throughput of the Doppler search against filter length: random banks of T taps at C2's geometry (D = 256, M = 8, N = 2^20),
the path the library chooses by itself and the two-pass path beside it.  Shows how the segment path runs out towards its limit
(T <= 2049: at least half of a 4096-point segment valid) and what the fall to the two-pass path beyond it amounts to.
usage: python tools/taps_sweep.py [T ...] [--also l,l,...]   (--also: segment lengths 2^l forced beside the chosen one)"""
import sys

import numpy as np

sys.path.insert(0, '.')
from pycusdr_amd.hostcpu import quiet_blas  # noqa: E402
quiet_blas()          # numpy's BLAS workers must not spend the container's CPU quota: a throttled host starves the device
from pycusdr_amd.mfbank import MFBank        # noqa: E402

log2N, D, M = 20, 256, 8
N = 1 << log2N
also = []
if '--also' in sys.argv:
    k = sys.argv.index('--also')
    also = [int(v) for v in sys.argv[k + 1].split(',')]
    del sys.argv[k:k + 2]
taps = [int(t) for t in sys.argv[1:]] or [16, 48, 80, 100, 160, 256, 384, 512, 768, 1025, 1536, 2049, 2050, 4096]
rs = np.random.RandomState(0)
x = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
shifts = np.sort(rs.choice(N, D, replace=False)).astype(np.int32)
bank = MFBank(log2N, D, M)
bank.set_shifts(shifts)
bank.upload(x)


def timed():
    bank.find_carrier()
    reps = 4
    prev = None
    for _ in range(12):           # until the device clock has settled (tools/ramp_probe.py)
        bank.timer_start()
        for _ in range(reps):
            bank.search_async()
        ms = bank.timer_stop() / reps
        if prev is not None and ms > 0.995 * prev:
            break
        prev = ms
    return ms


print('| taps | path chosen | valid per segment | ms per block | Msamples/s | two-pass ms | Msamples/s |')
print('|---|---|---|---|---|---|---|')
for T in taps:
    h = np.zeros((M, N), dtype=np.complex64)
    h[:, :T] = (rs.standard_normal((M, T)) + 1j * rs.standard_normal((M, T))).astype(np.complex64)
    masks = np.conj(np.fft.fft(h, axis=1)).astype(np.complex64)
    bank.set_search_path('auto')
    bank.set_filters(masks)
    info = bank.get_search_path()
    ms = timed()
    a = bank.get_scores()[:, 0].copy()
    bank.set_search_path('twopass')
    ms2 = timed()
    b = bank.get_scores()[:, 0]
    rel = float(np.abs(a - b).max() / b.max())
    assert rel < 1e-5, (T, rel)
    name = f"segment, L = {1 << info['log2L']}" if info['path'] == 'segment' else 'two-pass'
    v = info.get('valid_per_segment', 0) if info['path'] == 'segment' else '-'
    forced = ''
    for l in also:
        try:
            bank.set_search_path('segment', l)
            forced += f" L = {1 << l}: {timed():.2f} ms |"
        except ValueError:
            forced += f" L = {1 << l}: refused |"
    print(f"| {T} | {name} | {v} | {ms:.2f} | {(N - 1024) / ms / 1e3:.0f} | {ms2:.2f} | {(N - 1024) / ms2 / 1e3:.0f} |" + forced, flush=True)
bank.close()
