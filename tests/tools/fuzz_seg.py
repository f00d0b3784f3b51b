"""Randomised parity sweep of the search paths against the oracle at small sizes: random block length, bin count,
filter count, tap count and window position, segment length, sum_all on/off, noise bin, decomposition knobs, span
basis.  usage: python tests/tools/fuzz_seg.py [cases] [seed]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from oracle import mfbank_oracle as orc
from pycusdr_amd.mfbank import MFBank

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = 0.0
for case in range(cases):
    log2N = int(rs.randint(10, 17))
    N = 1 << log2N
    D = int(rs.randint(1, 40)) if rs.randint(0, 8) else int(rs.randint(200, 600))
    M = int(rs.choice([1, 2, 3, 5, 8, 13, 16, 17, 33, 64]))
    sum_all = bool(rs.randint(0, 2))
    doff = int(rs.randint(0, 2))
    lmax = min(13, log2N - 2)
    l = int(rs.randint(8, lmax + 1))
    L = 1 << l
    T = int(rs.randint(1, L // 2 + 2))
    start = int(rs.randint(0, N))
    rank = int(rs.randint(1, M + 1))
    h = np.zeros((M, N), dtype=np.complex128)
    idx = (start + np.arange(T)) % N
    basis = rs.standard_normal((rank, T)) + 1j * rs.standard_normal((rank, T))
    mix = rs.standard_normal((M, rank)) + 1j * rs.standard_normal((M, rank))
    h[:, idx] = mix @ basis                                 # rank-deficient banks exercise the span basis
    if M > 2 and rs.randint(0, 2):
        h[2] = -h[0]                                        # an exact negative: transformed once
    masks = np.fft.fft(h, axis=1).astype(np.complex64)
    x = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
    shifts = rs.randint(0, N, D + doff).astype(np.int32)
    bank = MFBank(log2N, D, M, sum_all_masks=sum_all, doppler_offset=doff)
    try:
        bank.set_filters(masks)
        bank.set_shifts(shifts)
        bank.upload(x)
        X = bank.get_spectrum()
        ref = orc.doppler_scores(X, masks, shifts, sum_all)
        tags = []

        def check(tag, idx_, ds):
            global worst
            err = float(np.abs(ds - ref).max() / ref.max())
            worst = max(worst, err)
            oidx, _ = orc.find_doppler_est(ds, D, doff, sum_all)
            ok = err < 1e-5 and (idx_ == oidx or (np.isnan(idx_) and np.isnan(oidx)))
            if not ok:
                print('FAIL', dict(case=case, log2N=log2N, D=D, M=M, sum_all=sum_all, doff=doff, l=l, T=T, start=start, rank=rank, tag=tag,
                                   err=err, idx=float(idx_), oidx=float(oidx), path=bank.get_search_path(), basis=bank.get_search_basis()))
                sys.exit(1)
        try:
            bank.set_search_path('segment', l, int(rs.choice([0, 1, 2, 3, 5, 6, 12, 24, 40, 64])), int(rs.randint(0, 17)))
            tags.append('segment')
        except ValueError:
            tags.append('twopass')                          # e.g. the random taps happened to be too long for this L
        for tag in list(tags) + (['span'] if sum_all and tags[0] == 'segment' else []) + ['twopass2']:
            if tag == 'span':
                bank.set_search_basis('span')
            if tag == 'twopass2':
                bank.set_search_basis('filters') if sum_all else None
                bank.set_search_path('twopass')
            for mode in ('transforms', 'energy') if tag != 'span' else ('transforms',):
                bank.set_search_mode(mode)
                idx_, met = bank.find_carrier()
                ds = bank.get_scores()
                check(tag + '/' + mode, idx_, ds)
        bank.set_search_mode('transforms')
        if tags[0] == 'segment':
            bank.set_search_path('segment', l)
            sh = int(shifts[-1])
            bank.demodulate(sh, 5, 50)
            xc = bank.get_xcorr()
            rxc = orc.demod_xcorr(X, masks, sh)
            e2 = float(np.abs(xc - rxc).max() / np.abs(rxc).max())
            worst = max(worst, e2)
            if e2 >= 1e-5:
                print('FAIL xcorr', dict(case=case, log2N=log2N, M=M, l=l, T=T, start=start, err=e2))
                sys.exit(1)
    finally:
        bank.close()
print(f'{cases} random cases ok, worst relative error {worst:.2e}')
