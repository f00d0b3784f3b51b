"""Scratch timing probe: find_carrier at a given geometry, sweeping tuning knobs."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from pycusdr_amd.hostcpu import quiet_blas  # noqa: E402
quiet_blas()          # numpy's BLAS workers must not spend the container's CPU quota: a throttled host starves the device
from pycusdr_amd.mfbank import MFBank


def parseval_scores(X, masks, shifts):
    """Sanity figure for the timed scores, no inverse transform: sum_n |IFFT(P)[n]|^2 = N sum_k |P[k]|^2, in float64,
    scaled by the 2^18 of the search (a property of the DFT, not the checker the parity tests use)."""
    w = (np.abs(np.asarray(masks, dtype=np.complex128)) ** 2).sum(axis=0)
    p = np.abs(np.asarray(X, dtype=np.complex128)) ** 2
    return np.array([len(p) * np.dot(np.roll(p, -int(s)), w) / 2.0 ** 18 for s in shifts])


log2N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
M = int(sys.argv[3]) if len(sys.argv) > 3 else 8
N = 1 << log2N
rs = np.random.RandomState(0)
x = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
masks = (rs.standard_normal((M, N)) + 1j * rs.standard_normal((M, N))).astype(np.complex64)
shifts = np.sort(rs.choice(N, D, replace=False)).astype(np.int32)
bank = MFBank(log2N, D, M)
bank.set_filters(masks); bank.set_shifts(shifts)
bank.upload(x)
X = bank.get_spectrum()
B_alg = 16.0 * D * M * N + 8.0 * N * (1 + M) + 16.0 * N + 4.0 * D
configs = [(128, 8, 64, 32), (128, 8, 64, 64), (128, 8, 64, 128), (128, 8, 32, 32), (128, 8, 16, 32), (128, 8, 32, 64), (128, 8, 64, 32), (128, 8, 16, 64)]
if len(sys.argv) > 4:
    configs = [tuple(int(v) for v in c.split(',')) for c in sys.argv[4:]]
for chunk, mpb, srb, js in configs:
    bank.set_tuning(chunk, mpb, srb, js)
    bank.find_carrier()
    reps = 3
    bank.timer_start()
    for _ in range(reps):
        bank.search_async()
    ms = bank.timer_stop() / reps
    bank.profile_enable(True)
    bank.search_async()
    cnt, tot = bank.profile_read()
    bank.profile_enable(False)
    print(f'chunk {chunk:3d} mpb {mpb} srb {srb:3d} js {js:2d}: {ms:8.3f} ms/block  {(N-1024)/ms/1e3:7.2f} Msamp/s  frac {B_alg/ms/1e-3/8e12:.3f}'
          f'  p1 {tot[0]:.2f} ms/{cnt[0]}  p2 {tot[1]:.2f} ms/{cnt[1]}', flush=True)
ds = bank.get_scores()[:, 0].astype(np.float64)
ref = parseval_scores(X, masks, shifts)
print('parseval rel err', np.abs(ds - ref).max() / ref.max())
bank.close()
