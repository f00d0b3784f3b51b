"""The CPU oracle against itself: identities and brute-force restatements of the reference's
kernel loops (the device stages have no fixtures in the reference -- see oracle header)."""
import numpy as np
import pytest

from oracle import mfbank_oracle as orc
from pycusdr_amd import config as cfg
from pycusdr_amd.demodulator.demodulator_base import doppler_bin_table


def _rc(rs, *s):
    return (rs.standard_normal(s) + 1j * rs.standard_normal(s)).astype(np.complex64)


def test_parseval_identity_of_doppler_scores():
    rs = np.random.RandomState(0)
    N, M, D = 4096, 4, 6
    X, masks = orc.forward_fft(_rc(rs, N)), _rc(rs, M, N)
    shifts = rs.randint(0, N, D)
    ds = orc.doppler_scores(X, masks, shifts, True)[:, 0]
    assert np.allclose(ds, orc.doppler_scores_parseval(X, masks, shifts), rtol=1e-10)
    per = orc.doppler_scores(X, masks, shifts, False)
    assert np.allclose(per.sum(axis=1), ds, rtol=1e-12)
    f32 = orc.doppler_scores(X, masks, shifts, True, dtype=np.complex64)[:, 0]
    assert np.abs(f32 - ds).max() / ds.max() < 1e-5


def test_shifted_product_is_the_kernel_loop():
    rs = np.random.RandomState(1)
    N = 64
    X, m = _rc(rs, N), _rc(rs, N)
    for s in (0, 5, N - 1):
        ref = np.array([X[(k + s) % N] * m[k] for k in range(N)])
        assert np.allclose(orc.shifted_product(X, m, s), ref, rtol=1e-6, atol=1e-6)


def test_doppler_table_matches_host_driver_and_wraps():
    conf = cfg.bench_config(doppCarrierSteps=64, blockSize=15)
    r = conf['Radios']['Rx']['UHF-H']
    N = 1 << 15
    t = orc.doppler_table(r['frequency_Hz'], r['frequencyOffset_Hz'], r['baud'], r['samplesPerSym'], 7500, 64, N)
    grid, hz, shifts, stx = doppler_bin_table(r, 7500, N)
    assert np.array_equal(t['shifts'], shifts) and np.array_equal(t['doppHzLUT'], hz) and t['doppOffsetIdx'] == stx == N // 4
    assert shifts.dtype == np.int32 and shifts.min() >= 0 and shifts.max() < N
    # negative IF offset: bins wrap into the upper half of the spectrum
    t2 = orc.doppler_table(437.3e6, -38400, 9600, 16, 7500, 8, N)
    assert np.all(t2['shifts'] > N // 2) and t2['doppOffsetIdx'] == N - N // 4
    t3 = orc.doppler_table(437.3e6, 38400, 9600, 16, 7500, 8, N, noise_measure_offset_Hz=-20000)
    assert t3['offset_count'] == 1 and len(t3['shifts']) == 9


def _pick_bruteforce(col, num, off):
    """Literal scalar restatement of one findDopplerEst thread in float64-free fp32."""
    v = [np.float32(0), np.float32(0)]
    ix = [0, 0]
    cur = 0
    for i in range(off, num + off):
        if col[i] > v[cur]:
            v[cur], ix[cur] = np.float32(col[i]), i
            cur = 1 if v[0] >= v[1] else 0
    return v, ix, cur


@pytest.mark.parametrize('seed', range(5))
def test_find_doppler_est_top2(seed):
    rs = np.random.RandomState(seed)
    D, M = 40, 4
    ds = rs.rand(D, M).astype(np.float32)
    idx, metric = orc.find_doppler_est(ds, D, 0, True)
    v, ix, _ = _pick_bruteforce(ds[:, 0], D, 0)
    order = np.argsort(-ds[:, 0])[:2]
    assert set(ix) == set(order.tolist())             # the two largest bins
    expect = (ix[0] * float(v[0]) + ix[1] * float(v[1])) / (float(v[0]) + float(v[1]))
    assert abs(float(idx) - expect) < 1e-4
    # quirk Q1: the weighted index may average non-adjacent bins
    ds2 = np.zeros((D, M), np.float32)
    ds2[3, 0], ds2[30, 0] = 2.0, 2.0
    idx2, _ = orc.find_doppler_est(ds2, D, 0, True)
    assert float(idx2) == 16.5
    # all-zero block -> NaN index -> host skips the block
    idx3, _ = orc.find_doppler_est(np.zeros((D, M), np.float32), D, 0, True)
    assert np.isnan(idx3) and orc.interpolate_doppler(idx3, np.arange(D), np.arange(D, dtype=float)) is None
    # per-mask mode averages the per-mask estimates
    idx4, _ = orc.find_doppler_est(ds, D, 0, False)
    per = [orc.find_doppler_est(ds[:, m:m + 1], D, 0, True)[0] for m in range(M)]
    assert abs(float(idx4) - float(np.mean(per))) < 1e-4


def test_interpolate_doppler_rounds_like_the_host():
    shifts = np.array([100, 110, 121, 133], dtype=np.int32)
    hz = np.array([1.0, 2.0, 3.0, 4.0])
    r = orc.interpolate_doppler(np.float32(1.5), shifts, hz, 0.5)
    assert r['low'] == 1 and r['high'] == 2 and r['dopplerIdxlast'] == int(np.round(110 + 11 * 0.5)) and r['freqOffset'] == 2.0
    r = orc.interpolate_doppler(np.float32(2.0), shifts, hz)
    assert r['low'] == r['high'] == 2 and r['dopplerIdxlast'] == 121


def _centres_bruteforce(xc, spSym, offset, W, op):
    """Scalar transcription of the findCentres thread body (cuda_kernels.cu:78-146), fp32."""
    f = np.float32
    M, N = xc.shape
    S = int(N / spSym)
    sp, off = f(spSym), f(offset)
    sym, cen, mag = np.zeros(S, np.int32), np.zeros(S, np.int32), np.zeros(S, np.float32)
    for x in range(S):
        base = f(np.float64(f(x)) * np.float64(sp) - np.float64(f(W // 2)))
        a = int(np.trunc(f(base + off)))
        mx = a + W
        oc = int(np.trunc(off))
        if a < 0:
            oc -= a
            a = 0
        mx = min(mx, N) - a
        best, bi, bk = f(0), -1, -1
        for m in range(M):
            for k in range(mx):
                z = xc[m, a + k]
                if op == 0:
                    t = f(np.float64(z.real) * np.float64(z.real) + np.float64(f(z.imag * z.imag)))
                elif op == 1:
                    t = f(abs(z.real))
                else:
                    t = f(abs(z.imag))
                if t > best:
                    best, bi, bk = t, m, k
        sym[x], mag[x] = bi, best
        cen[x] = int(np.trunc(f(f(base + f(bk)) + f(oc))))
    return sym, cen, mag


@pytest.mark.parametrize('op', [0, 1, 2])
def test_find_centres_vectorised_equals_scalar(op):
    rs = np.random.RandomState(3 + op)
    M, N = 4, 2048
    xc = _rc(rs, M, N)
    xc[:, 700:760] = 0            # a stretch with no maximum at all -> symbol -1
    for spSym, offset in ((16.0, 3.7), (15.93, 0.2), (8.01, 6.9)):
        a = orc.find_centres(xc, spSym, offset, 7, op)
        b = _centres_bruteforce(xc, spSym, offset, 7, op)
        for u, v in zip(a, b):
            assert np.array_equal(u, v)
        assert (a[0] == -1).any()


def test_code_rate_host_and_window():
    k_off, k_len = orc.code_rate_window(1 << 16, 16)
    assert k_off == int(65536 / 17.6) and k_off + k_len == int(65536 / 14.4)
    sp, co = orc.code_rate_host(4096, 0.5, 1 << 16)
    assert sp == 16.0 and abs(co - (-0.5 / np.pi * 8 + 15)) < 1e-6
    sp, co = orc.code_rate_host(0, 0.1, 1 << 16)
    assert sp == 10.0
    env = 1 + np.cos(2 * np.pi * np.arange(4096) / 16 + 0.7)
    k, arg, val = orc.code_rate_and_phase(env, *orc.code_rate_window(4096, 16))
    assert k == 256 and abs(arg - 0.7) < 1e-9


def test_sync_correlate_and_candidates():
    rs = np.random.RandomState(9)
    tmpl_bits = rs.randint(0, 2, 32)
    mask = np.flipud(tmpl_bits * 2 - 1)
    bits = rs.randint(0, 2, 500)
    bits[100:132] = tmpl_bits
    sc = orc.sync_correlate(bits, mask)
    idx, start = orc.header_candidates(sc, tmpl_bits.sum(), 0, len(mask))
    assert 100 in start
    assert sc[131] == tmpl_bits.sum()


def test_snr_band_selection():
    N = 1024
    X = np.ones(N, np.complex64)
    X[250:262] = 10
    shifts = np.array([250, 261])
    snr = orc.compute_snr(X, shifts, 0, 1, 5, N)
    sig = np.mean(np.abs(X[245:266]))
    assert abs(snr - 20 * np.log10(sig / 1.0 - 1)) < 1e-9


def test_frequency_domain_bank_equals_time_domain_matched_filter():
    """First-principles cross-check of A4/A5/A9: multiplying the shifted spectrum by conj(fft(template))
    and inverse-transforming (unnormalised, as cuFFT does) IS N times the circular cross-correlation of
    the de-rotated signal with the template: y[n] = N * sum_l conj(t[l]) * xd[n+l],
    xd[n] = x[n] e^{-2 pi i s n/N}."""
    rs = np.random.RandomState(21)
    N, Lt, s = 512, 24, 37
    x = _rc(rs, N).astype(np.complex128)
    t = _rc(rs, Lt).astype(np.complex128)
    mask = np.conj(np.fft.fft(t, N))
    y = orc.demod_xcorr(np.fft.fft(x), mask[None, :], s)[0]
    n = np.arange(N)
    xd = x * np.exp(-2j * np.pi * s * n / N)                  # Doppler removed in the time domain
    direct = np.array([np.sum(np.conj(t) * xd[(k + np.arange(Lt)) % N]) for k in range(N)])
    assert np.allclose(y, N * direct, rtol=1e-9, atol=1e-7)
    # and the per-bin score is the energy of that correlation, scaled by N^2 / 2^18
    score = orc.doppler_scores(np.fft.fft(x), mask[None, :], [s], True)[0, 0]
    assert np.isclose(score, N * N * np.sum(np.abs(direct) ** 2) / 262144.0, rtol=1e-9)


# ---- the oracle's HOST restatements against the reference-run fixtures G15-G17 (tests/golden/ref_goldens_host.npz) -----------
def _hg():
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_goldens_host.npz'), allow_pickle=False)
    return {k.replace('__', '/'): z[k] for k in z.files}


def test_oracle_host_functions_equal_reference_runs():
    """doppler_table (DB:130-165), interpolate_doppler (DB:609-622), compute_snr (DB:635-667) and code_rate_host (DB:733-752) of the
    oracle against what the reference's own code computed under the recording fake (legacy reading of the rate arithmetic)."""
    import json
    hg = _hg()
    for name in sorted({k.split('/')[1] for k in hg if k.startswith('g15/')}):
        conf = json.loads(str(hg[f'g15/{name}/conf']))
        r = conf['Radios']['Rx']['UHF-H']
        t = orc.doppler_table(r['frequency_Hz'], r['frequencyOffset_Hz'], r['baud'], r['samplesPerSym'], conf['Radios']['rangeRateMax'],
                              r['doppCarrierSteps'], int(hg[f'g15/{name}/Nfft']), r.get('noise_measure_offset_Hz', False))
        assert np.array_equal(t['shifts'], hg[f'g15/{name}/doppCyperSymNorm']) and np.array_equal(t['doppHzLUT'], hg[f'g15/{name}/doppHzLUT'])
        assert t['doppOffsetIdx'] == int(hg[f'g15/{name}/doppOffsetIdx']) and t['offset_count'] == int(hg[f'g15/{name}/doppIdxArrayOffset'])
    for name in ('bench_b15_d64', 'zero_if', 'noise_pos', 'cc11xx_b17_s128'):
        p = f'g16/{name}/'
        N = int(hg[f'g15/{name}/Nfft'])
        rs = np.random.RandomState(int(hg[p + 'spectrum_seed']))
        X = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
        X[N // 4 - 40:N // 4 + 40] *= 25
        X[:24] *= 9
        X[-24:] *= 9
        shifts, hz = hg[f'g15/{name}/doppCyperSymNorm'], hg[f'g15/{name}/doppHzLUT']
        for i, pick in enumerate(hg[p + 'pick']):
            got = orc.interpolate_doppler(pick, shifts, hz, float(hg[f'g15/{name}/centreFreqOffset']))
            if np.isnan(pick):
                assert got is None and hg[p + 'dopplerIdxlast'][i] == 0
                continue
            assert got['dopplerIdxlast'] == int(hg[p + 'dopplerIdxlast'][i]) and got['freqOffset'] == hg[p + 'freqOffset'][i]
            with np.errstate(all='ignore'):
                snr = orc.compute_snr(X, shifts, got['low'], got['high'], 5, N)
            assert np.array_equal(np.float64(snr), hg[p + 'SNR'][i], equal_nan=True)
    for name in ('bench_b15_d64', 'cc11xx_b17_s128'):
        p = f'g17/{name}/'
        N = int(hg[f'g15/{name}/Nfft'])
        for i in range(0, len(hg[p + 'k']), 7):
            sp, off = orc.code_rate_host(hg[p + 'k'][i], hg[p + 'arg'][i], N)
            assert sp == hg[p + 'legacy/spSym'][i] and np.array_equal(np.float64(off), hg[p + 'legacy/codeOffset'][i], equal_nan=True)
