"""Timing probe of the segment search at a given geometry: every segment length L and a few
decompositions, against the two-pass path, with a Parseval sanity check of the scores.
usage: python tools/seg_probe.py [log2N] [D] [protocol] [l-list] [wpc-list]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from pycusdr_amd.hostcpu import quiet_blas  # noqa: E402
quiet_blas()          # numpy's BLAS workers must not spend the container's CPU quota: a throttled host starves the device
from pycusdr_amd import config as cfg, signals as sg
from pycusdr_amd.mfbank import MFBank
from pycusdr_amd.protocol import loadProtocol
from pycusdr_amd.demodulator.demodulator_base import doppler_bin_table


def parseval_scores(X, masks, shifts):
    """Sanity figure for the timed scores, no inverse transform: sum_n |IFFT(P)[n]|^2 = N sum_k |P[k]|^2, in float64,
    scaled by the 2^18 of the search (a property of the DFT, not the checker the parity tests use)."""
    w = (np.abs(np.asarray(masks, dtype=np.complex128)) ** 2).sum(axis=0)
    p = np.abs(np.asarray(X, dtype=np.complex128)) ** 2
    return np.array([len(p) * np.dot(np.roll(p, -int(s)), w) / 2.0 ** 18 for s in shifts])


log2N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
name = sys.argv[3] if len(sys.argv) > 3 else 'bench_GMSK'
ls = [int(v) for v in sys.argv[4].split(',')] if len(sys.argv) > 4 else [8, 9, 10, 11, 12]
wpcs = [int(v) for v in sys.argv[5].split(',')] if len(sys.argv) > 5 else [2]
N = 1 << log2N
if name == 'CC11xx':
    conf = cfg.cc11xx_config(blockSize=log2N, doppCarrierSteps=D)
    sps, ms = 128, 3
else:
    conf = cfg.bench_config(name, blockSize=log2N, doppCarrierSteps=D)
    sps, ms = 16, (5 if name == 'bench_BPSK' else 3)
_, _, shifts, _ = doppler_bin_table(conf['Radios']['Rx']['UHF-H'], conf['Radios']['rangeRateMax'], N)
M, masks = loadProtocol(name)(conf=conf).get_filter(N, sps, ms)
x = sg.s1_stream(1, N, 1 << 10, 'GMSK', snr_db=10.0, seed=1)[:N]
if '--zeros' in sys.argv:      # power/clock experiment: the same instruction stream on all-zero data
    x = np.zeros(N, dtype=np.complex64)
bank = MFBank(log2N, D, M)
bank.set_filters(masks)
bank.set_shifts(shifts)
bank.upload(x)
X = bank.get_spectrum()
ref = parseval_scores(X, masks, shifts)
print('auto:', bank.get_search_path(), flush=True)


def run(tag):
    bank.find_carrier()
    reps = 5
    # The device needs ~30 ms of work after an idle spell (handle construction, filter analysis, set_search_path) to settle
    # its clock (tools/ramp_probe.py: 1.92 -> 1.60 ms per search at C2): time groups of `reps` until one is no faster than
    # the one before it, and report the last.
    prev = None
    for _ in range(12):
        bank.timer_start()
        for _ in range(reps):
            bank.search_async()
        ms_ = bank.timer_stop() / reps
        if prev is not None and ms_ > 0.995 * prev:
            break
        prev = ms_
    ds = bank.get_scores()[:, 0].astype(np.float64)
    err = np.abs(ds - ref).max() / max(ref.max(), 1e-30)
    print(f'{tag:34s} {ms_:8.3f} ms/block  {(N - 1024) / ms_ / 1e3:8.2f} Msamp/s   parseval rel err {err:.2e}', flush=True)


for l in ls:
    for w in wpcs:
        try:
            bank.set_search_path('segment', l, w)
        except ValueError as e:
            print(f'L=2^{l}: refused ({e})')
            break
        info = bank.get_search_path()
        run(f"segment L=2^{l} V={info['valid_per_segment']} wpc={w}")
if '--no-twopass' not in sys.argv:
    bank.set_search_path('twopass')
    run('twopass')
bank.close()
