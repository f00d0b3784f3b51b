"""Decoder through its default (HIP) sync correlator against the reference KATs (fixtures G4)."""
import numpy as np
import pytest

from oracle import mfbank_oracle as orc
from pycusdr_amd import config as cfg
from pycusdr_amd.decoder import Decoder, _hip_correlator
from pycusdr_amd.mfbank import sync_correlate
from pycusdr_amd.protocol import loadProtocol

from test_decoder_host import CASES, run_kat

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('key,pname,conf', CASES)
def test_findframes_kats_on_hip_correlator(goldens, key, pname, conf):
    run_kat(goldens, key, pname, conf, None)       # None -> default HIP path


def test_batched_streams_s_1024():
    """1024 synthetic bit streams, header planted every 4000 bits."""
    p = loadProtocol('bench_GMSK')(conf=cfg.bench_config())
    mask = p.get_mask()
    hdr = ((np.flipud(mask) + 1) // 2).astype(np.uint8)
    rs = np.random.RandomState(2)
    B, L = 1024, 16384
    bits = rs.randint(0, 2, (B, L)).astype(np.uint8)
    for s in range(0, L - 128, 4000):
        bits[:, s:s + 128] = hdr
    out = sync_correlate(bits, mask)
    assert out.shape == (B, L + 127) and out.dtype == np.int32
    for b in (0, 511, 1023):
        assert np.array_equal(out[b], orc.sync_correlate(bits[b], mask))
    hits = np.where(out[7] >= p.numOnesHeader)[0] - 127
    assert set(range(0, L - 128, 4000)).issubset(set(hits.tolist()))


def test_edge_cases_and_errors():
    assert np.array_equal(sync_correlate(np.array([1], np.uint8), np.array([1, -1, 1])), [1, -1, 1])
    with pytest.raises(ValueError):
        sync_correlate(np.array([0, 2, 1]), np.array([1, -1]))          # not a bit stream
    with pytest.raises(ValueError):
        sync_correlate(np.zeros(0, np.uint8), np.array([1, -1]))        # empty input
    with pytest.raises(ValueError):
        sync_correlate(np.array([0, 1], np.uint8), np.array([0.5, 1]))  # non-integer template
    f = _hip_correlator(np.array([1., 0., 1., 1.]), np.array([1., -1.]))
    assert np.array_equal(f, np.convolve([1, 0, 1, 1], [1, -1]))


def test_sync_find_equals_where_of_convolve():
    from pycusdr_amd.mfbank import sync_find
    rs = np.random.RandomState(12)
    for B, L, T, thr in ((1, 70000, 64, 20), (5, 3000, 128, 24), (3, 900, 16, 4), (2, 5000, 32, -40)):
        bits = rs.randint(0, 2, (B, L)).astype(np.uint8)
        tmpl = (rs.randint(0, 2, T) * 2 - 1).astype(np.int8)
        got = sync_find(bits, tmpl, thr, max_hits=8)        # small max_hits forces the regrow path
        for b in range(B):
            ref = orc.sync_correlate(bits[b], tmpl)
            idx = np.where(ref >= thr)[0]
            assert np.array_equal(got[b][0], idx) and np.array_equal(got[b][1], ref[idx])
    i1, s1 = sync_find(bits[0], tmpl, 3.5)                   # 1-D input, fractional threshold
    ref = orc.sync_correlate(bits[0], tmpl)
    assert np.array_equal(i1, np.where(ref >= 3.5)[0]) and np.array_equal(s1, ref[ref >= 3.5])


def test_sync_find_multi_equals_separate_searches():
    """Several templates of different lengths in one call: small inputs take the packed single-copy route, large ones
    the direct copies; both must equal np.where(np.convolve(...) >= thr) per template and stream."""
    from pycusdr_amd.mfbank import sync_find_multi
    rs = np.random.RandomState(21)
    for B, L, Ts, thrs in ((1, 2304, (128, 16), (20, 6)), (1, 67584, (64, 32, 7), (18, 10, 5)), (40, 60000, (64, 200), (22, 40)),
                           (3, 50, (64, 5), (-10, 2)), (2, 1024, (1, 4096), (1, 90))):
        bits = rs.randint(0, 2, (B, L)).astype(np.uint8)
        tmpls = [(rs.randint(0, 2, T) * 2 - 1).astype(np.int8) for T in Ts]
        got = sync_find_multi(bits, tmpls, thrs, max_hits=4)          # small max_hits forces the regrow path
        assert len(got) == len(Ts)
        for k, (tm, thr) in enumerate(zip(tmpls, thrs)):
            for b in range(B):
                ref = orc.sync_correlate(bits[b], tm)
                idx = np.where(ref >= thr)[0]
                assert np.array_equal(got[k][b][0], idx) and np.array_equal(got[k][b][1], ref[idx]), (B, L, k, b)
    # 1-D float bit stream, float templates, fractional thresholds: what Decoder.findFrames passes
    fb = bits[0].astype(np.float64)
    (i0, s0), (i1, s1) = sync_find_multi(fb, [tmpls[0].astype(float), tmpls[1].astype(float)], (0.5, 80.5))
    r0, r1 = orc.sync_correlate(bits[0], tmpls[0]), orc.sync_correlate(bits[0], tmpls[1])
    assert np.array_equal(i0, np.where(r0 >= 0.5)[0]) and np.array_equal(s1, r1[r1 >= 80.5]) and np.array_equal(i1, np.where(r1 >= 80.5)[0])
    with pytest.raises(ValueError):
        sync_find_multi(np.array([0., 0.5, 1.]), [tmpls[0]], (1,))
    with pytest.raises(ValueError):
        sync_find_multi(bits, tmpls, (1,))


@pytest.mark.parametrize('name', ['inside', 'across', 'overflow', 'two'])
def test_flags_mode_kats_on_hip_correlator(goldens, name):
    """The reference's FLAGS-mode findFrames KATs (fixture G11) with both correlations on the GPU."""
    from test_pins_round2 import run_flags_kat
    assert run_flags_kat(goldens, name, None) >= 1


@pytest.mark.parametrize('tag', ['r1_vs_d1', 'r2_vs_d2_16k', 'r1_shifted'])
def test_custom_xcorr_on_hip_matches_reference(goldens, tag):
    """N4: the soft combiner's alignment cross-correlation on the HIP transforms (mfb_xcorr) against the
    fixture recorded from the reference's customXCorr on its own unit-test streams: peak index exact,
    magnitudes within 1e-5 of the peak, complex values against the oracle restatement."""
    from oracle import mfbank_oracle as orc
    from pycusdr_amd.mfbank import customXCorr
    a, b = goldens[f'g13/{tag}/a'].astype(np.float64), goldens[f'g13/{tag}/b'].astype(np.float64)
    r = customXCorr(a, b)
    assert r.dtype == np.complex64 and len(r) == len(a)
    mag = np.abs(r)
    ref = goldens[f'g13/{tag}/abs_c64']
    assert int(np.argmax(mag)) == int(goldens[f'g13/{tag}/top15_idx'][0])
    assert np.abs(mag - ref).max() / ref.max() < 1e-5
    full = orc.custom_xcorr(a, b)
    assert np.abs(r - full).max() / np.abs(full).max() < 1e-5
    # N given explicitly, b longer than N (fft truncates), and a non-power-of-two N refused
    r2 = customXCorr(a[:5000], b, N=4096)
    f2 = orc.custom_xcorr(a[:5000], b, 4096)
    assert np.abs(r2 - f2).max() / np.abs(f2).max() < 1e-5
    with pytest.raises(ValueError):
        customXCorr(a, b, N=5000)


@pytest.mark.parametrize('L,T,B', [(67584, 64, 5), (1000, 64, 3), (4097, 128, 2), (777, 16, 4), (20001, 100, 2), (63, 64, 1), (130, 200, 2)])
def test_sync_find_packed_equals_convolve(L, T, B):
    """Packed sync correlation (np.packbits layout, XOR/AND + popcount on 64-bit windows): positions and scores equal
    np.where(np.convolve(bits, template) >= threshold), stream by stream, for any length, tap count (zeros included) and
    with garbage in the padding bits of the last byte."""
    from pycusdr_amd.mfbank import sync_find_packed, sync_pinned_buffer
    rs = np.random.RandomState(L + T)
    bits = rs.randint(0, 2, (B, L)).astype(np.uint8)
    tmpl = rs.choice([-1, 1], T).astype(np.int8)
    if T > 20:
        tmpl[rs.randint(0, T, 3)] = 0
    header = ((tmpl[::-1] + 1) // 2).astype(np.uint8)
    for pos in range(5, L - T, max(T + 30, L // 7)):
        bits[:, pos:pos + T] = header
        bits[0, pos + 3] ^= 1
    packed = np.packbits(bits, axis=1)
    if L % 8:
        packed[:, -1] |= (1 << (8 - L % 8)) - 1                      # padding bits set: they are not part of the stream
    thr = int((tmpl == 1).sum()) - 3
    got = sync_find_packed(packed, L, tmpl, thr)
    for b in range(B):
        sc = np.convolve(bits[b].astype(np.int64), tmpl.astype(np.int64))
        idx = np.where(sc >= thr)[0]
        assert np.array_equal(got[b][0], idx) and np.array_equal(got[b][1], sc[idx]), (b, len(idx), len(got[b][0]))
    assert sum(len(g[0]) for g in got) > 0 or L < T
    # a low threshold (many hits, more than the first guess of room), rows wider than needed, the page-locked staging
    thr2 = -2
    stage = sync_pinned_buffer(B * (packed.shape[1] + 5))
    wide = stage[:B * (packed.shape[1] + 5)].reshape(B, packed.shape[1] + 5)
    wide[:] = 255
    wide[:, :packed.shape[1]] = packed
    got2, ms = sync_find_packed(wide, L, tmpl, thr2, max_total=16, timing=True)
    for b in range(B):
        sc = np.convolve(bits[b].astype(np.int64), tmpl.astype(np.int64))
        idx = np.where(sc >= thr2)[0]
        assert np.array_equal(got2[b][0], idx) and np.array_equal(got2[b][1], sc[idx])
    assert ms >= 0.0
    cnt, fidx, fsc = sync_find_packed(wide, L, tmpl, thr2, flat=True)          # the library's own flat form
    assert np.array_equal(cnt, [len(g[0]) for g in got2])
    assert np.array_equal(fidx, np.concatenate([g[0] for g in got2])) and np.array_equal(fsc, np.concatenate([g[1] for g in got2]))
    with pytest.raises(ValueError):
        sync_find_packed(packed, L, np.array([2, 1, -1], np.int8), 1)      # taps outside {-1, 0, +1}


def test_sync_finder_object_begin_end_and_growth():
    """The decoder's finder object: templates resident, begin / end, more hits than room (the finder is rebuilt with enough
    of it), streams longer than its first allocation, long streams (multi-pass form) -- always np.convolve's hits."""
    from pycusdr_amd.mfbank import SyncFinder
    rs = np.random.RandomState(9)
    t1 = rs.choice([-1, 1], 64).astype(np.int8)
    t2 = rs.choice([-1, 1], 16).astype(np.int8)
    thr = (int((t1 == 1).sum()) - 6, 3)                      # the second threshold is low: thousands of hits
    f = SyncFinder((t1, t2), thr, max_hits=4, max_bits=256)
    try:
        for L in (100, 2300, 70000, 300):
            bits = rs.randint(0, 2, L).astype(np.uint8)
            hdr = ((t1[::-1] + 1) // 2).astype(np.uint8)
            if L > 200:
                bits[50:114] = hdr
            f.begin(bits.astype(np.float64))                  # the decoder hands over float bits
            got = f.end()
            for (idx, sc), t, th in zip(got, (t1, t2), thr):
                ref = np.convolve(bits.astype(np.int64), t.astype(np.int64))
                want = np.where(ref >= th)[0]
                assert np.array_equal(idx, want) and np.array_equal(sc, ref[want]), (L, len(want), len(idx))
        assert f.max_hits > 4
    finally:
        f.close()
