"""GPU parity tests of the HIP kernels against the CPU oracle, through the C ABI (ctypes)."""
import numpy as np
import pytest

from oracle import mfbank_oracle as orc

pytestmark = pytest.mark.gpu


def _rand_c64(rs, *shape):
    return (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)


@pytest.mark.parametrize('log2N', [10, 11, 12, 13, 15, 16, 17, 18])
def test_forward_fft_matches_numpy(log2N):
    from pycusdr_amd.mfbank import MFBank
    rs = np.random.RandomState(log2N)
    N = 1 << log2N
    x = _rand_c64(rs, N)
    bank = MFBank(log2N, 4, 2)
    try:
        bank.upload(x)
        X = bank.get_spectrum()
    finally:
        bank.close()
    ref = np.fft.fft(x.astype(np.complex128))
    err = np.abs(X - ref).max() / np.abs(ref).max()
    assert err < 2e-6, err      # fp32 FFT; tolerance stated: 2e-6 of the spectrum peak


@pytest.mark.parametrize('log2N,D,M,sum_all', [(10, 5, 2, True), (12, 8, 4, True), (13, 3, 8, False),
                                                (15, 8, 8, True), (16, 32, 8, True), (17, 4, 3, False),
                                                (12, 4, 64, False), (12, 3, 33, True)])
def test_doppler_scores_match_oracle(log2N, D, M, sum_all):
    from pycusdr_amd.mfbank import MFBank
    rs = np.random.RandomState(100 + log2N)
    N = 1 << log2N
    x = _rand_c64(rs, N)
    masks = _rand_c64(rs, M, N)
    shifts = rs.randint(0, N, D).astype(np.int32)
    shifts[0] = 0
    shifts[-1] = N - 1
    bank = MFBank(log2N, D, M, sum_all_masks=sum_all)
    try:
        bank.set_filters(masks)
        bank.set_shifts(shifts)
        bank.upload(x)
        idx, metric = bank.find_carrier()
        ds = bank.get_scores()
        X = bank.get_spectrum()
        # chunking must not change anything (bit-reproducible fixed-order reductions)
        bank.set_tuning(doppler_chunk=1, masks_per_block=1)
        idx2, metric2 = bank.find_carrier()
        ds2 = bank.get_scores()
    finally:
        bank.close()
    ref = orc.doppler_scores(X, masks, shifts, sum_all)
    rel = np.abs(ds - ref).max() / ref.max()
    assert rel < 1e-5, rel                          # north_star: 1e-5 on correlation magnitudes
    assert np.array_equal(ds, ds2) and idx == idx2 and metric == metric2
    if sum_all:
        assert np.all(ds[:, 1:] == 0)               # only column 0 populated (reference quirk Q2)
    # the pick is exact fp32 arithmetic on the device's own doppSum
    oidx, ometric = orc.find_doppler_est(ds, D, 0, sum_all)
    assert idx == oidx
    assert abs(float(metric) - float(ometric)) <= 2e-6 * abs(float(ometric)) + 1e-6


def test_pick_with_noise_bin_and_external_scores():
    from pycusdr_amd.mfbank import MFBank
    rs = np.random.RandomState(5)
    log2N, D, M = 12, 6, 4
    N = 1 << log2N
    bank = MFBank(log2N, D, M, doppler_offset=1)
    try:
        bank.set_filters(_rand_c64(rs, M, N))
        bank.set_shifts(rs.randint(0, N, D + 1))
        bank.upload(_rand_c64(rs, N))
        idx, metric = bank.find_carrier()
        ds = bank.get_scores()
    finally:
        bank.close()
    oidx, ometric = orc.find_doppler_est(ds, D, 1, True)
    assert idx == oidx
    assert abs(float(metric) - float(ometric)) <= 2e-6 * abs(float(ometric)) + 1e-6


@pytest.mark.parametrize('log2N,M,sps', [(12, 4, 8), (15, 8, 16), (16, 8, 16)])
def test_demod_stage_matches_oracle(log2N, M, sps):
    from pycusdr_amd.mfbank import MFBank
    rs = np.random.RandomState(200 + log2N)
    N = 1 << log2N
    # a signal with symbol-rate structure so the envelope spectrum has a clear line
    nsym = N // sps
    symv = rs.randint(0, 2, nsym) * 2.0 - 1.0
    pulse = np.hanning(sps)
    x = (np.repeat(symv, sps) * np.tile(pulse, nsym)).astype(np.complex64) * np.exp(1j * 0.3)
    x = (x + 0.05 * _rand_c64(rs, N)).astype(np.complex64)
    masks = np.zeros((M, N), dtype=np.complex64)
    for m in range(M):
        t = rs.standard_normal(sps) + 1j * rs.standard_normal(sps) * 0.1
        masks[m] = np.conj(np.fft.fft(t, N)).astype(np.complex64)
    shift = 3
    k_off, k_len = orc.code_rate_window(N, sps)
    bank = MFBank(log2N, 2, M, window_width=7)
    try:
        bank.set_filters(masks)
        bank.set_shifts([0, 1])
        bank.upload(x)
        X = bank.get_spectrum()
        k, arg, val = bank.demodulate(shift, k_off, k_len)
        xc = bank.get_xcorr()
        env = bank.get_envelope()
        spSym, codeOffset = orc.code_rate_host(k, arg, N)
        S = int(N / spSym)
        sym, cen, mag = bank.find_centres(np.float32(spSym), np.float32(codeOffset), 0, S)
    finally:
        bank.close()
    ref_xc = orc.demod_xcorr(X, masks, shift)
    scale = np.abs(ref_xc).max()
    assert np.abs(xc - ref_xc).max() / scale < 5e-6
    ref_env = orc.envelope(xc)
    assert np.abs(env - ref_env).max() / ref_env.max() < 1e-6
    ok, oarg, oval = orc.code_rate_and_phase(env, k_off, k_len)
    assert int(k) == ok
    assert abs(float(arg) - oarg) < 1e-4
    assert abs(float(val) - oval) / oval < 1e-5
    # symbol decisions: bit-exact against the oracle on the device's own matched-filter outputs
    osym, ocen, omag = orc.find_centres(xc, spSym, codeOffset, 7, 0)
    assert np.array_equal(sym, osym)
    assert np.array_equal(cen, ocen)
    assert np.array_equal(mag.view(np.uint32), omag.view(np.uint32))


@pytest.mark.parametrize('B,L,T', [(1, 100, 16), (3, 5000, 64), (4, 70000, 128), (2, 31, 64)])
def test_sync_correlate_exact(B, L, T):
    from pycusdr_amd.mfbank import sync_correlate
    rs = np.random.RandomState(B * 1000 + T)
    bits = rs.randint(0, 2, (B, L)).astype(np.uint8)
    tmpl = (rs.randint(0, 2, T) * 2 - 1).astype(np.int8)
    out = sync_correlate(bits, tmpl)
    for b in range(B):
        assert np.array_equal(out[b], orc.sync_correlate(bits[b], tmpl))


@pytest.mark.parametrize('sum_all', [True, False])
def test_duplicate_and_negated_filters_are_transformed_once_but_reported_for_all(sum_all):
    """Bank rows [A, B, -A, A, C]: the search transforms 3 unique rows; doppSum must still be what the
    oracle gets from all 5 (bit-identical energies for the copies)."""
    from pycusdr_amd.mfbank import MFBank
    rs = np.random.RandomState(77)
    log2N, D = 13, 9
    N = 1 << log2N
    A, B, Cc = _rand_c64(rs, N), _rand_c64(rs, N), _rand_c64(rs, N)
    masks = np.stack([A, B, -A, A, Cc]).astype(np.complex64)
    shifts = rs.randint(0, N, D).astype(np.int32)
    bank = MFBank(log2N, D, 5, sum_all_masks=sum_all)
    try:
        bank.set_filters(masks)
        bank.set_shifts(shifts)
        bank.upload(_rand_c64(rs, N))
        idx, metric = bank.find_carrier()
        ds = bank.get_scores()
        X = bank.get_spectrum()
        k, arg, val = bank.demodulate(int(shifts[0]), 100, 200)     # demodulation keeps all 5 rows
        xc = bank.get_xcorr()
    finally:
        bank.close()
    ref = orc.doppler_scores(X, masks, shifts, sum_all)
    assert np.abs(ds - ref).max() / ref.max() < 1e-5
    if not sum_all:
        assert np.array_equal(ds[:, 0], ds[:, 2]) and np.array_equal(ds[:, 0], ds[:, 3])
    assert idx == orc.find_doppler_est(ds, D, 0, sum_all)[0]
    assert np.array_equal(xc[2], -xc[0]) and np.array_equal(xc[3], xc[0])


def test_find_centres_randomised_parameters_bit_exact():
    """Symbol-centre search over random symbol rates, phases, window widths and operations: symbol
    index, centre and fp32 magnitude must equal the oracle's bit for bit (same matched-filter outputs)."""
    from pycusdr_amd.mfbank import MFBank
    rs = np.random.RandomState(2024)
    log2N = 13
    N = 1 << log2N
    for W in (3, 7, 9):
        M = int(rs.choice([2, 5, 8]))
        bank = MFBank(log2N, 2, M, window_width=W)
        try:
            bank.set_filters(_rand_c64(rs, M, N))
            bank.set_shifts([0, 1])
            x = _rand_c64(rs, N)
            x[3000:3300] = 0                       # a dead stretch: symbols with no maximum (-1)
            bank.upload(x)
            bank.demodulate(int(rs.randint(0, N)), 10, 50)
            xc = bank.get_xcorr()
            for _ in range(6):
                spSym = float(rs.uniform(4.0, 40.0))
                offset = float(rs.uniform(0.0, spSym))
                op = int(rs.randint(0, 3))
                S = int(N / spSym)
                sym, cen, mag = bank.find_centres(np.float32(spSym), np.float32(offset), op, S)
                osym, ocen, omag = orc.find_centres(xc, spSym, offset, W, op)
                assert np.array_equal(sym, osym), (W, M, spSym, offset, op)
                assert np.array_equal(cen, ocen)
                assert np.array_equal(mag.view(np.uint32), omag.view(np.uint32))
        finally:
            bank.close()


def test_pick_randomised_tables_exact():
    """Doppler pick on externally supplied score tables (the multi-GPU path hands the all-reduced table
    to mfb_pick): index bit-exact, metric to 1 ulp-ish, for random D, M, noise-bin offsets, both modes,
    including ties and all-equal tables."""
    import torch
    from pycusdr_amd.mfbank import MFBank
    rs = np.random.RandomState(99)
    for M, sum_all in ((1, True), (4, True), (8, False), (32, False), (3, False)):
        bank = MFBank(10, 4, M, sum_all_masks=sum_all)
        try:
            for trial in range(6):
                D = int(rs.randint(2, 700))
                off = int(rs.randint(0, 2))
                tab = rs.rand(D + off, M).astype(np.float32) + 0.01
                if trial == 4:
                    tab[:] = 1.0                                   # every bin ties
                if trial == 5:
                    tab[off + D // 2] = tab[off + D // 3]          # two equal maxima candidates
                t = torch.from_numpy(tab).cuda()
                idx, metric = bank.pick(t.data_ptr(), num=D, offset=off)
                oidx, ometric = orc.find_doppler_est(tab, D, off, sum_all)
                assert idx == oidx, (M, sum_all, D, off, trial)
                assert abs(float(metric) - float(ometric)) <= 3e-6 * abs(float(ometric)) + 1e-6
        finally:
            bank.close()


def test_pick_closed_form_and_scan_agree_with_the_reference_scan():
    """k_pick's closed form (records / first occurrences) and its sequential fallback against the oracle's restatement
    of the reference scan, on the shapes that decide which slot a maximum ends up in: ramps, descents, peaks, plateaus,
    ties with the running maximum early and late, quantised tables full of ties, zeros, negatives, NaN, inf, a single
    positive value, nothing positive, and columns longer than the staging buffer."""
    import torch
    from pycusdr_amd.mfbank import MFBank
    rs = np.random.RandomState(123)

    def tables():
        for D in (1, 2, 3, 63, 64, 65, 128, 256, 700, 4096, 4097, 6000):
            base = rs.rand(D).astype(np.float32) + 0.01
            yield 'random', base
            yield 'ramp', np.sort(base)
            yield 'descent', np.sort(base)[::-1].copy()
            peak = np.concatenate((np.sort(base[:D // 2]), np.sort(base[D // 2:])[::-1]))
            yield 'peak', peak
            q = np.round(base * 7).astype(np.float32)                   # heavy ties, zeros included
            yield 'quantised', q
            t = base.copy()
            if D > 4:
                t[D // 2] = t[:D // 2].max()                            # equals the running maximum when it arrives
                yield 'tie-with-running-max', t
                t2 = base.copy()
                t2[-1] = t2.max()                                       # a second copy of the global maximum at the end
                yield 'tie-with-global-max', t2
                t3 = base.copy()
                t3[D // 3] = 0.0
                t3[D // 4] = -1.0
                t3[D // 5] = np.nan
                yield 'zero-negative-nan', t3
                t4 = base.copy()
                t4[D // 2] = np.inf
                yield 'inf', t4
            one = np.zeros(D, np.float32)
            one[D // 2] = 3.0
            yield 'single', one
            yield 'nothing', np.zeros(D, np.float32)

    bank = MFBank(10, 4, 1, sum_all_masks=True)
    bank3 = MFBank(10, 4, 3, sum_all_masks=False)
    try:
        for name, col in tables():
            for off in (0, 1):
                D = len(col) - off
                if D < 1:
                    continue
                tab = np.ascontiguousarray(col[:, None])
                t = torch.from_numpy(tab).cuda()
                with np.errstate(all='ignore'):
                    oidx, ometric = orc.find_doppler_est(tab, D, off, True)
                idx, metric = bank.pick(t.data_ptr(), num=D, offset=off)
                assert np.array_equal(np.float32(idx), np.float32(oidx), equal_nan=True), (name, len(col), off, idx, oidx)
                if np.isfinite(ometric):
                    assert abs(float(metric) - float(ometric)) <= 3e-6 * abs(float(ometric)) + 1e-6, (name, len(col), off)
        for trial in range(12):                                          # several columns, the mean over filters
            D = int(rs.randint(2, 5000))
            tab = (np.round(rs.rand(D, 3) * (50 if trial % 2 else 1e6)) / 7).astype(np.float32)
            t = torch.from_numpy(tab).cuda()
            with np.errstate(all='ignore'):
                oidx, ometric = orc.find_doppler_est(tab, D, 0, False)
            idx, metric = bank3.pick(t.data_ptr(), num=D, offset=0)
            assert np.array_equal(np.float32(idx), np.float32(oidx), equal_nan=True), (trial, D)
    finally:
        bank.close()
        bank3.close()


def test_spectrum_windows_from_the_host_mirror():
    """The first spectrum read of a handle turns on the host mirror (small blocks): later reads, served from page-locked
    host memory, must equal the device spectrum of the block uploaded last -- whole, windowed and wrapped."""
    from pycusdr_amd.mfbank import MFBank
    log2N = 13
    N = 1 << log2N
    rs = np.random.RandomState(9)
    bank = MFBank(log2N, 4, 2)
    try:
        for block in range(4):
            x = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
            bank.upload(x)
            ref = np.fft.fft(x.astype(np.complex128))
            full = bank.get_spectrum()                    # block 0: fetched from the device; later: from the mirror
            assert np.abs(full - ref).max() / np.abs(ref).max() < 2e-6
            for start, count in ((0, 7), (N - 3, 10), (N // 2 - 5, 11), (-4, 9), (N - 1, 1), (5, 0)):
                w = bank.get_spectrum(start, count)
                assert np.array_equal(w, np.take(full, np.arange(start, start + count), mode='wrap'))
        xd = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
        bank.upload(xd)
        bank.upload(x)                                     # two uploads back to back: the mirror follows the last one
        assert np.array_equal(bank.get_spectrum(), full)
    finally:
        bank.close()


@pytest.mark.parametrize('sum_all', [True, False])
def test_scores_and_demod_outputs_against_the_vendor_fft(sum_all):
    """The reference's formulation evaluated with the vendor FFT in fp32 on the device (torch.fft = rocFFT): per-filter
    row sums and the matched-filter outputs at one shift, against both search paths."""
    torch = pytest.importorskip('torch')
    from pycusdr_amd import config as cfg
    from pycusdr_amd.mfbank import MFBank
    from pycusdr_amd.protocol import loadProtocol
    log2N, D = 15, 24
    N = 1 << log2N
    rs = np.random.RandomState(3)
    M, masks = loadProtocol('bench_FSK')(conf=cfg.bench_config('bench_FSK', blockSize=log2N)).get_filter(N, 16, 3)
    x = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
    shifts = rs.randint(0, N, D).astype(np.int32)
    bank = MFBank(log2N, D, M, sum_all_masks=sum_all)
    try:
        bank.set_filters(masks)
        bank.set_shifts(shifts)
        bank.upload(x)
        dev = torch.device('cuda', 0)
        X = torch.from_numpy(bank.get_spectrum()).to(dev)
        H = torch.from_numpy(np.ascontiguousarray(masks)).to(dev)
        ref = np.zeros((D, M))
        for j in range(D):
            y = torch.fft.ifft(torch.roll(X, -int(shifts[j]))[None, :] * H, dim=1, norm='forward')
            ref[j] = ((y.real.double() ** 2 + y.imag.double() ** 2).sum(dim=1) / 2 ** 18).cpu().numpy()
        want = ref.sum(axis=1) if sum_all else ref
        for path in ('segment', 'twopass'):
            bank.set_search_path(path)
            bank.find_carrier()
            ds = bank.get_scores()
            got = ds[:, 0] if sum_all else ds
            assert np.abs(got - want).max() / want.max() < 1e-5, path
            k_off, k_len = int(N / (1.1 * 16)), int(N / (0.9 * 16)) - int(N / (1.1 * 16))
            kstar, arg, mag = bank.demodulate(int(shifts[5]), k_off, k_len)
            yt = torch.fft.ifft(torch.roll(X, -int(shifts[5]))[None, :] * H, dim=1, norm='forward')
            y = yt.cpu().numpy()
            xc = bank.get_xcorr()
            assert np.abs(xc - y).max() / np.abs(y).max() < 1e-5, path
            # A10 with the vendor FFT: envelope, its real-input spectrum, windowed argmax, phase
            env = (yt.real ** 2 + yt.imag ** 2).sum(dim=0)
            assert np.abs(bank.get_envelope() - env.cpu().numpy()).max() / float(env.max()) < 1e-5
            P = torch.fft.rfft(env)[k_off:k_off + k_len]
            p2 = (P.real.double() ** 2 + P.imag.double() ** 2).cpu().numpy()
            assert int(kstar) == k_off + int(p2.argmax())
            assert abs(float(mag) - p2.max()) / p2.max() < 1e-4
            ref_arg = float(torch.atan2(P.imag[int(p2.argmax())], P.real[int(p2.argmax())]))
            assert abs(np.angle(np.exp(1j * (float(arg) - ref_arg)))) < 1e-3
    finally:
        bank.close()
