"""GPU parity tests of the single-pass overlap-save path (k_seg) against the CPU oracle and against the
two-pass path of the same handle, through the C ABI.  Every segment length, both synchronisation
flavours (wave-local for L <= 1024, workgroup barrier for 2048/4096), odd decompositions, partial last
segments, filters whose support sits anywhere on the circle, duplicate filters, the noise bin."""
import numpy as np
import pytest

from oracle import mfbank_oracle as orc
from pycusdr_amd import config as cfg
from pycusdr_amd.mfbank import MFBank
from pycusdr_amd.protocol import loadProtocol

pytestmark = pytest.mark.gpu


def _rc(rs, *s):
    return (rs.standard_normal(s) + 1j * rs.standard_normal(s)).astype(np.complex64)


def _bank_masks(name, log2N, sps=16, ms=3):
    conf = (cfg.cc11xx_config(blockSize=log2N) if name == 'CC11xx' else cfg.bench_config(name, blockSize=log2N))
    return loadProtocol(name)(conf=conf).get_filter(1 << log2N, sps, ms)


def _short_masks(rs, M, N, T, start):
    """Random T-tap impulse responses placed at [start, start+T) on the circle, as spectra."""
    h = np.zeros((M, N), dtype=np.complex128)
    idx = (start + np.arange(T)) % N
    h[:, idx] = rs.standard_normal((M, T)) + 1j * rs.standard_normal((M, T))
    return (np.fft.fft(h, axis=1)).astype(np.complex64)


@pytest.mark.parametrize('log2L', [8, 9, 10, 11, 12, 13])
@pytest.mark.parametrize('sum_all', [True, False])
def test_segment_scores_match_oracle_and_twopass(log2L, sum_all):
    log2N, D = 14, 11
    N = 1 << log2N
    rs = np.random.RandomState(log2L)
    M, masks = _bank_masks('bench_GMSK', log2N)
    x = _rc(rs, N)
    shifts = rs.randint(0, N, D).astype(np.int32)
    shifts[0], shifts[1], shifts[-1] = 0, N // 4, N - 1
    bank = MFBank(log2N, D, M, sum_all_masks=sum_all)
    try:
        bank.set_filters(masks)
        bank.set_shifts(shifts)
        assert bank.get_search_path()['path'] == 'segment' and bank.get_search_path()['taps'] == 48
        bank.set_search_path('segment', log2L)
        info = bank.get_search_path()
        L = 1 << log2L
        nt = L // (32 if L == 2048 else 16)      # transform lanes: 16 points per lane; 32 in the wave-local 2048-point kernel
        assert info['log2L'] == log2L and info['valid_per_segment'] == ((L - 47) // nt) * nt   # whole register slots
        bank.upload(x)
        idx, metric = bank.find_carrier()
        ds = bank.get_scores()
        idx_b, metric_b = bank.find_carrier()
        assert np.array_equal(ds, bank.get_scores()) and idx == idx_b        # bit-reproducible
        X = bank.get_spectrum()
        bank.set_search_path('twopass')
        assert bank.get_search_path()['path'] == 'twopass'
        idx2, _ = bank.find_carrier()
        ds2 = bank.get_scores()
    finally:
        bank.close()
    ref = orc.doppler_scores(X, masks, shifts, sum_all)
    assert np.abs(ds - ref).max() / ref.max() < 1e-5          # north_star tolerance on magnitudes
    assert np.abs(ds - ds2).max() / ds2.max() < 2e-6          # the two device paths agree to fp32 rounding
    oidx, _ = orc.find_doppler_est(ds, D, 0, sum_all)
    assert idx == oidx
    if sum_all:
        assert np.all(ds[:, 1:] == 0)


@pytest.mark.parametrize('log2L,wpc,fpp', [(8, 1, 1), (8, 3, 3), (9, 2, 5), (10, 1, 8), (10, 3, 2), (11, 2, 3), (12, 1, 7), (12, 3, 16),
                                            (8, 64, 2), (9, 48, 0), (12, 64, 4), (13, 2, 3), (13, 32, 8)])
def test_segment_decompositions_do_not_change_results(log2L, wpc, fpp):
    """Workgroups per CU and filters per pass only regroup the work: scores stay within rounding of the
    default decomposition (the partial sums are regrouped) and the pick stays put."""
    log2N, D, M = 15, 7, 8
    N = 1 << log2N
    rs = np.random.RandomState(77)
    masks = _short_masks(rs, M, N, 61, N - 30)
    x = _rc(rs, N)
    shifts = rs.randint(0, N, D).astype(np.int32)
    bank = MFBank(log2N, D, M, sum_all_masks=False)
    try:
        bank.set_filters(masks)
        bank.set_shifts(shifts)
        bank.upload(x)
        bank.set_search_path('segment', log2L)
        bank.find_carrier()
        base = bank.get_scores()
        bank.set_search_path('segment', log2L, wpc, fpp)
        bank.find_carrier()
        ds = bank.get_scores()
        X = bank.get_spectrum()
        with pytest.raises(ValueError):
            bank.set_search_path('segment', log2L, 65, fpp)          # at most 64 workgroups per CU in the grid
    finally:
        bank.close()
    ref = orc.doppler_scores(X, masks, shifts, False)
    assert np.abs(ds - ref).max() / ref.max() < 1e-5
    assert np.abs(ds - base).max() / base.max() < 1e-6


@pytest.mark.parametrize('log2L,T,start', [(8, 1, 0), (8, 128, 5), (9, 200, 16000), (10, 512, 16384 - 100), (11, 777, 3),
                                           (12, 2048, 9000), (12, 33, 16383), (13, 4097, 5), (13, 2500, 14000)])
def test_segment_any_support_window_scores_and_xcorr(log2L, T, start):
    """Taps anywhere on the circle (wrapping included), from 1 tap up to L/2: scores (REDUCE mode) and the
    natural-order matched-filter outputs (STORE mode) against the oracle."""
    log2N, D, M = 14, 5, 3
    N = 1 << log2N
    rs = np.random.RandomState(T)
    masks = _short_masks(rs, M, N, T, start)
    x = _rc(rs, N)
    shifts = np.array([0, 1, N // 3, N - 2, 4097], dtype=np.int32)
    bank = MFBank(log2N, D, M, sum_all_masks=False)
    try:
        bank.set_filters(masks)
        bank.set_search_path('segment', log2L)
        info = bank.get_search_path()
        assert info['path'] == 'segment' and info['taps'] == T
        bank.set_shifts(shifts)
        bank.upload(x)
        bank.find_carrier()
        ds = bank.get_scores()
        X = bank.get_spectrum()
        bank.demodulate(int(shifts[2]), 10, 100)
        xc = bank.get_xcorr()
    finally:
        bank.close()
    ref = orc.doppler_scores(X, masks, shifts, False)
    assert np.abs(ds - ref).max() / ref.max() < 1e-5
    ref_xc = orc.demod_xcorr(X, masks, int(shifts[2]))
    assert np.abs(xc - ref_xc).max() / np.abs(ref_xc).max() < 1e-5


@pytest.mark.parametrize('name,sps,ms,log2N,log2L', [('bench_BPSK', 16, 5, 16, 0), ('bench_BPSK', 16, 5, 15, 9),
                                                     ('CC11xx', 128, 3, 16, 0), ('CC11xx', 128, 3, 16, 12),
                                                     ('bench_FSK', 16, 3, 13, 0), ('bench_GFSK', 16, 3, 16, 8)])
def test_segment_shipped_banks(name, sps, ms, log2N, log2L):
    """The shipped filter generators (BPSK: 32 filters, 16 unique up to sign; CC11xx: 384 taps) with the
    automatic and a forced segment length; noise-reference bin included."""
    N = 1 << log2N
    rs = np.random.RandomState(log2N + ms)
    M, masks = _bank_masks(name, log2N, sps, ms)
    D = 6
    x = _rc(rs, N)
    shifts = rs.randint(0, N, D + 1).astype(np.int32)
    bank = MFBank(log2N, D, M, sum_all_masks=True, doppler_offset=1)
    try:
        bank.set_filters(masks)
        if log2L:
            bank.set_search_path('segment', log2L)
        info = bank.get_search_path()
        assert info['path'] == 'segment'
        bank.set_shifts(shifts)
        bank.upload(x)
        idx, metric = bank.find_carrier()
        ds = bank.get_scores()
        X = bank.get_spectrum()
        bank.demodulate(int(shifts[3]), 10, 100)
        xc = bank.get_xcorr()
    finally:
        bank.close()
    ref = orc.doppler_scores(X, masks, shifts, True)
    assert np.abs(ds - ref).max() / ref.max() < 1e-5
    oidx, ometric = orc.find_doppler_est(ds, D, 1, True)
    assert idx == oidx
    ref_xc = orc.demod_xcorr(X, masks, int(shifts[3]))
    assert np.abs(xc - ref_xc).max() / np.abs(ref_xc).max() < 1e-5


def test_segment_refused_for_long_filters_and_small_blocks():
    rs = np.random.RandomState(3)
    log2N, M = 12, 2
    N = 1 << log2N
    bank = MFBank(log2N, 3, M)
    try:
        bank.set_filters(_rc(rs, M, N))                       # white spectra: support = N
        info = bank.get_search_path()
        assert info['path'] == 'twopass' and info['taps'] == N
        with pytest.raises(ValueError):
            bank.set_search_path('segment')
        assert bank.get_search_path()['path'] == 'twopass'     # previous setting stays in force
        with pytest.raises(ValueError):
            bank.set_search_path('segment', 14)
        with pytest.raises(ValueError):
            bank.set_search_path('segment', 13)                # 8192-point segments need N >= 2^14
        bank.set_filters(_short_masks(rs, M, N, 700, 0))       # 700 taps in a 4096-sample block: L would be 2048 > N/4
        assert bank.get_search_path()['path'] == 'twopass'
        bank.set_filters(_short_masks(rs, M, N, 20, 7))
        assert bank.get_search_path() == dict(path='segment', log2L=8, taps=20, valid_per_segment=224, segments=19)
        bank.set_filters(_short_masks(rs, M, N, 90, 4000))       # 90 taps: 256-point segments still win (V = 160)
        assert bank.get_search_path()['log2L'] == 8 and bank.get_search_path()['valid_per_segment'] == 160
    finally:
        bank.close()


def test_zero_and_single_tap_filters():
    """An all-zero filter row and a pure delay: energy sums are 0 and N * sum|x|^2 / 2^18."""
    log2N, M = 13, 3
    N = 1 << log2N
    rs = np.random.RandomState(9)
    h = np.zeros((M, N), dtype=np.complex128)
    h[1, 100] = 1.0
    h[2, 103] = 2.0
    masks = np.fft.fft(h, axis=1).astype(np.complex64)
    x = _rc(rs, N)
    bank = MFBank(log2N, 2, M, sum_all_masks=False)
    try:
        bank.set_filters(masks)
        assert bank.get_search_path()['path'] == 'segment' and bank.get_search_path()['taps'] == 4
        bank.set_shifts([5, 900])
        bank.upload(x)
        bank.find_carrier()
        ds = bank.get_scores()
    finally:
        bank.close()
    e = float(N) * N * np.sum(np.abs(x.astype(np.complex128)) ** 2) / 262144.0
    assert np.all(ds[:, 0] == 0)
    assert np.allclose(ds[:, 1], e, rtol=1e-5) and np.allclose(ds[:, 2], 4 * e, rtol=1e-5)


@pytest.mark.parametrize('name,sps,ms,rank', [('bench_GMSK', 16, 3, 6), ('bench_FSK', 16, 3, 4), ('bench_BPSK', 16, 5, 5),
                                              ('CC11xx', 128, 3, 4)])
def test_span_basis_search_equals_full_bank(name, sps, ms, rank):
    """Opt-in span basis: the SUM_ALL search over an orthogonalised basis of the bank's span (rank filters instead
    of M) gives the reference's doppSum -- against the oracle's FULL bank (1e-5) and against the default search of
    the same handle (fp32 rounding); the pick is the same; demodulation still uses all M filters."""
    log2N, D = 16, 9
    N = 1 << log2N
    rs = np.random.RandomState(rank)
    M, masks = _bank_masks(name, log2N, sps, ms)
    x = _rc(rs, N)
    shifts = rs.randint(0, N, D).astype(np.int32)
    bank = MFBank(log2N, D, M, sum_all_masks=True)
    try:
        bank.set_filters(masks)
        bank.set_shifts(shifts)
        bank.upload(x)
        assert bank.get_search_basis()[0] == 'filters'
        idx0, _ = bank.find_carrier()
        ds0 = bank.get_scores()
        bank.set_search_basis('span')
        assert bank.get_search_basis() == ('span', rank)
        idx1, _ = bank.find_carrier()
        ds1 = bank.get_scores()
        X = bank.get_spectrum()
        bank.demodulate(int(shifts[2]), 10, 100)
        xc = bank.get_xcorr()
        bank.set_search_path('twopass')                       # the request stays pending off the segment path
        assert bank.get_search_basis()[0] == 'filters'
        bank.set_search_path('segment')
        assert bank.get_search_basis() == ('span', rank)
        bank.set_search_basis('filters')
        assert bank.get_search_basis()[0] == 'filters'
    finally:
        bank.close()
    ref = orc.doppler_scores(X, masks, shifts, True)
    assert np.abs(ds1 - ref).max() / ref.max() < 1e-5
    assert np.abs(ds1 - ds0).max() / ds0.max() < 2e-6
    # white-noise input: all bins score alike, so the weighted top-2 index moves in its last bits with the scores'
    assert int(idx0) == int(idx1) and abs(float(idx0) - float(idx1)) < 1e-5 * max(1.0, abs(float(idx0)))
    assert np.all(ds1[:, 1:] == 0)
    ref_xc = orc.demod_xcorr(X, masks, int(shifts[2]))
    assert xc.shape == (M, N) and np.abs(xc - ref_xc).max() / np.abs(ref_xc).max() < 1e-5


def test_span_basis_refused_without_sum_all():
    bank = MFBank(12, 3, 2, sum_all_masks=False)
    try:
        with pytest.raises(RuntimeError):
            bank.set_search_basis('span')
    finally:
        bank.close()


@pytest.mark.parametrize('name,log2N,D,Doff,rect', [('bench_GMSK', 17, 96, 0, None), ('bench_GMSK', 17, 96, 0, '16,1'), ('bench_BPSK', 16, 40, 2, '8,1'),
                                                     ('CC11xx', 17, 64, 0, None), ('CC11xx', 17, 64, 0, '8,1'), ('bench_GMSK', 15, 9, 1, '5,2')])
def test_the_shift_on_the_filters_side_equals_the_shift_on_the_samples(tmp_path, name, log2N, D, Doff, rect):
    """Round 6's form of the segment search -- a segment transformed once for several bins, the Doppler shift on the filters' side
    (k_segf, seg_kernels.hpp) -- against the form of rounds 2-5 -- every (bin, segment) mixed in time and transformed (k_seg) -- on the
    same seeded block: two processes (MFB_SEG_FSM is read once per process), the same table to fp32 rounding, the same pick; 256-point
    segments (8 and 16 unique filters, with and without noise-reference rows, bins that do not fill a rectangle) and the wave-local
    2048-point ones (CC11xx, 384 taps).  ``rect`` = MFB_SEG_FSM_RECT: a wave's rectangle of bins x slots; None = the library's own
    choice, which for problems this small shrinks the rectangle to keep the launch wide (down to one bin: the forward transform is
    then not shared, only the mixing multiply is gone)."""
    import os
    import subprocess
    import sys
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'children', 'fsm_child.py')
    res = {}
    for fsm in ('0', '1'):
        out = str(tmp_path / f'fsm{fsm}.npz')
        env = dict(os.environ, MFB_SEG_FSM=fsm)
        env.pop('MFB_SEG_FSM_RECT', None)
        if rect:
            env['MFB_SEG_FSM_RECT'] = rect
        subprocess.run([sys.executable, child, name, str(log2N), str(D), str(Doff), out], check=True, env=env, timeout=300)
        res[fsm] = np.load(out)
    a, b = res['0'], res['1']
    assert int(a['filter_side']) == 0 and int(b['filter_side']) == 1 and int(b['bins_per_forward']) >= (2 if rect else 1)
    assert int(a['log2L']) == int(b['log2L']) == (11 if name == 'CC11xx' else 8)
    sa, sb = a['scores'].astype(np.float64), b['scores'].astype(np.float64)
    assert sa.shape == sb.shape == (D + Doff, sa.shape[1])
    assert np.abs(sa - sb).max() / sa.max() < 2e-6            # two device formulations: fp32 rounding apart (the oracle gate is 1e-5)
    assert not np.array_equal(sa, sb)                          # ... and really two formulations
    assert abs(float(a['pick'][0]) - float(b['pick'][0])) < 1e-3 and np.all(sb[:, 1:] == 0)


@pytest.mark.parametrize('log2L,T', [(8, 40), (8, 80), (11, 300)])
def test_sum_all_total_once_per_bin_with_filters_that_count_differently(log2L, T):
    """SUM_ALL searches take the Parseval half of a segment's energy once per bin from a host table of sum_f w_f |G_f|^2 (segf_body,
    SUMQ), w_f = how often k_finalize counts row f relative to row 0.  A bank whose rows count 3, 1, 2 and 1 times (exact copies and an
    exact negative are transformed once): the SUM_ALL column against the oracle's sum over all seven filters, and against the sum of
    the per-filter columns of a handle without SUM_ALL (which keeps the per-filter form), both shifts-on-the-filters searches."""
    log2N, D = 15, 37
    N = 1 << log2N
    rs = np.random.RandomState(100 + T)
    base = _short_masks(rs, 4, N, T, 777)
    masks = np.stack([base[0], base[1], -base[0], base[2], base[2], base[0], base[3]]).astype(np.complex64)      # rows count 3, 1, 2, 1
    x = _rc(rs, N)
    shifts = rs.randint(0, N, D).astype(np.int32)
    tables = {}
    for sum_all in (True, False):
        bank = MFBank(log2N, D, 7, sum_all_masks=sum_all)
        try:
            bank.set_filters(masks)
            bank.set_shifts(shifts)
            bank.set_search_path('segment', log2L)
            assert bank.get_info()[2] == 4                                   # four rows transformed
            info = bank.get_search_info()
            assert info['filter_side'] and info['bins_per_forward'] >= 1
            bank.upload(x)
            bank.find_carrier()
            tables[sum_all] = bank.get_scores().astype(np.float64)
            X = bank.get_spectrum()
        finally:
            bank.close()
    ref = orc.doppler_scores(X, masks, shifts, True)[:, 0]
    assert np.abs(tables[True][:, 0] - ref).max() <= 1e-5 * ref.max()
    assert np.abs(tables[True][:, 0] - tables[False].sum(axis=1)).max() <= 2e-6 * ref.max()
    assert np.all(tables[True][:, 1:] == 0)
