"""The host arithmetic inside the device-calling methods, pinned by REFERENCE-RUN fixtures G15-G18
(tests/golden/ref_goldens_host.npz, written by tests/golden/make_golden_host.py: the reference's own constructor,
``uploadAndFindCarrier`` and ``demodulate`` executed under a recording fake of its driver, device results injected where
its ``memcpy_dtoh`` calls deliver them).  Here this repo's host driver runs over ``ReplayBank`` with the same values
injected at the same points; everything it returns, and everything it would launch the device with, must equal the
reference's -- bit for bit.

Where numpy's scalar promotion changes the reference's result (DB:623, 735, 745: float32 element x Python scalar) the
fixtures hold both readings; the build follows ``legacy`` (numpy < 2: float64), the only one the reference -- which uses
``np.float`` / ``np.int`` -- ever ran under unshimmed.  The tests assert the legacy reading and state what NEP 50 changes.
"""
import json
import os

import numpy as np
import pytest

from pycusdr_amd.demodulator import STX, UHF
from pycusdr_amd.demodulator.demodulator_base import doppler_bin_table
from pycusdr_amd.protocol import loadProtocol
import pycusdr_amd.demodulator.demodulator_base as dbm

from replay_bank import ReplayBank

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def hg():
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'ref_goldens_host.npz'), allow_pickle=False)
    return {k.replace('__', '/'): z[k] for k in z.files}


def cases(hg):
    return sorted({k.split('/')[1] for k in hg if k.startswith('g15/')})


def build(hg, name, monkeypatch):
    """This repo's Demodulator for fixture case ``name`` over a ReplayBank."""
    conf = json.loads(str(hg[f'g15/{name}/conf']))
    backend = {'UHF': UHF, 'STX': STX}[str(hg[f'g15/{name}/backend'])]
    proto = loadProtocol(str(hg[f'g15/{name}/protocol']))(conf=conf)
    with monkeypatch.context() as m:
        m.setattr(dbm, 'MFBank', ReplayBank)
        d = backend.Demodulator(conf, proto, 'UHF-H')
    assert type(d.bank) is ReplayBank and not d._one_call
    return d


def spectrum(N, seed):
    rs = np.random.RandomState(seed)
    X = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
    X[N // 4 - 40:N // 4 + 40] *= 25
    X[:24] *= 9
    X[-24:] *= 9
    return X


def same(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


CASES15 = ['bench_b15_d32', 'bench_b15_d64', 'bench_b15_d256', 'bench_b15_d1024', 'bench_b15_d2048', 'bench_b16_d32', 'bench_b16_d64',
           'bench_b16_d256', 'bench_b16_d1024', 'bench_b16_d2048', 'bench_b20_d32', 'bench_b20_d64', 'bench_b20_d256', 'bench_b20_d1024',
           'bench_b20_d2048', 'bench_b20_d1024_rr30000', 'bench_b20_d2048_rr60000', 'noise_neg', 'noise_pos', 'zero_if', 'neg_if',
           'bpsk_b15', 'cc11xx_b17_s128', 'cc11xx_b16_s16', 'stx_b14']


def test_fixture_cases_are_the_ones_listed(hg):
    assert cases(hg) == sorted(CASES15)


@pytest.mark.parametrize('name', CASES15)
def test_doppler_table_equals_reference_constructor(hg, name):
    """A1 (DB:130-165): bin positions, Hz lookup, integer shifts (negatives wrapped), STX shift -- every float64 bit."""
    conf = json.loads(str(hg[f'g15/{name}/conf']))
    N = int(hg[f'g15/{name}/Nfft'])
    grid, hz, shifts, stx = doppler_bin_table(conf['Radios']['Rx']['UHF-H'], conf['Radios']['rangeRateMax'], N)
    p = f'g15/{name}/'
    assert same(grid, hg[p + 'doppIdxNorm']) and grid.dtype == np.float64
    assert same(hz, hg[p + 'doppHzLUT'])
    assert same(shifts, hg[p + 'doppCyperSymNorm']) and shifts.dtype == hg[p + 'doppCyperSymNorm'].dtype == np.int32
    assert stx == int(hg[p + 'doppOffsetIdx'])


@pytest.mark.parametrize('name', ['bench_b15_d64', 'bench_b20_d256', 'noise_neg', 'noise_pos', 'zero_if', 'neg_if', 'bpsk_b15',
                                  'cc11xx_b17_s128', 'cc11xx_b16_s16', 'stx_b14'])
def test_constructor_scalars_equal_reference(hg, name, monkeypatch):
    """What else the constructor derives (DB:75-128, 508-512): clamp, rate-search window, overlap window, thresholds."""
    d = build(hg, name, monkeypatch)
    for f in ('doppIdxArrayLen', 'doppIdxArrayOffset', 'spsymMin', 'sampleRate', 'sigOverlapWin', 'windowWidthOffset',
              'symbol_check_match_threshold', 'overlapOffset', 'symbol_check_error_threshold', 'codeRateAndPhaseOffsetLow',
              'codeRateAndPhaseOffsetHigh', 'num_masks', 'Nfft', 'SUM_ALL_MASKS_PYTHON', 'CODE_SEARCH_MASK_OFFSET', 'num_dopplers',
              'centreFreqOffset'):
        assert same(getattr(d, f), hg[f'g15/{name}/{f}']), f
    assert same(d.bank.shifts, hg[f'g15/{name}/doppCyperSymNorm'])          # the table the device gets (DB:221)
    # the reference's capacity for symbols: int(N / spsymMin) (DB:468-472)
    assert int(d.Nfft / d.spsymMin) == int(hg[f'g15/{name}/centres_capacity'])


@pytest.mark.parametrize('name', ['bench_b15_d64', 'zero_if', 'neg_if', 'noise_neg', 'noise_pos', 'cc11xx_b16_s16', 'cc11xx_b17_s128',
                                  'bench_b20_d256'])
def test_find_carrier_host_half_equals_reference(hg, name, monkeypatch):
    """A7 host half + A8 (DB:604-667) on injected findDopplerEst results: integer and fractional indices, both ends of the
    table, bins either side of 0 Hz (shift interpolation across the wrap), noise-reference bin in front, NaN (block skipped)."""
    d = build(hg, name, monkeypatch)
    p = f'g16/{name}/'
    d.bank.X = spectrum(d.Nfft, int(hg[p + 'spectrum_seed']))
    for i, (pick, metric) in enumerate(zip(hg[p + 'pick'], hg[p + 'metric'])):
        d.bank.pick = (pick, metric)
        with np.errstate(all='ignore'):
            fo, sd, clipped, snr = d.uploadAndFindCarrier(d.get_signalBufferHostPointer())
        assert same(np.float64(fo), hg[p + 'freqOffset'][i]), (i, pick)
        assert same(np.float64(snr), hg[p + 'SNR'][i]), (i, pick, snr, hg[p + 'SNR'][i])
        assert int(d.dopplerIdxlast) == int(hg[p + 'dopplerIdxlast'][i])
        # DB:623 under NEP 50 is a float32 expression; ours is float64 (legacy).  N is a power of two and the sample rate an
        # integer below 2^24, so the float64 value rounds to exactly the float32 one
        assert same(np.float32(sd), np.float32(hg[p + 'sdev_Hz_nep50'][i]))
        assert len(clipped) == 0
    assert np.isnan(hg[p + 'pick'][-1]) and hg[p + 'dopplerIdxlast'][-1] == 0 and hg[p + 'freqOffset'][-1] == 0


def run_rate(d, k, arg):
    d.bank.triple = (k, arg, 1.0)
    d.bank.calls.clear()
    d.dopplerIdxlast = 0
    spSym, off = d.findCodeRateAndPhaseGPU()
    res = d.cudaFindCentres(spSym, off)
    fc = [c for c in d.bank.calls if c[0] == 'find_centres'][0]
    return spSym, off, fc, res


@pytest.mark.parametrize('name', ['bench_b15_d64', 'bench_b20_d256', 'cc11xx_b17_s128'])
def test_rate_phase_arithmetic_equals_reference(hg, name, monkeypatch):
    """A10 host half + the launch of A11 (DB:733-752, 994-1006) over every k* of the rate window x eleven phases: samples per
    symbol, code offset (wrap of negatives), the clamp, the symbol count and the two float32 launch arguments."""
    d = build(hg, name, monkeypatch)
    cap = int(d.Nfft / d.spsymMin)
    d.bank.sym, d.bank.cen, d.bank.mag = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.float32)
    p = f'g17/{name}/'
    L = {f: hg[p + 'legacy/' + f] for f in ('spSym', 'codeOffset', 'spSymF', 'phaseF', 'count')}
    ks, args = hg[p + 'k'], hg[p + 'arg']
    step = 1 if len(ks) < 6000 else 3
    for i in range(0, len(ks), step):
        spSym, off, fc, res = run_rate(d, ks[i], args[i])
        assert same(np.float64(spSym), L['spSym'][i]) and same(np.float64(off), L['codeOffset'][i]), (ks[i], args[i])
        assert same(fc[1], L['spSymF'][i]) and same(fc[2], L['phaseF'][i]) and fc[4] == L['count'][i] == len(res[0])
        assert fc[3] == int(hg[p + 'op'])
    # what NEP 50 (numpy >= 2) changes in the reference, for the record: never the samples-per-symbol argument, but the
    # phase argument and the symbol count in a fair share of the cases
    Nq = {f: hg[p + 'nep50/' + f] for f in ('spSymF', 'phaseF', 'count')}
    assert same(Nq['spSymF'], L['spSymF'])
    assert 0.05 < np.mean(Nq['count'] != L['count']) < 0.5 and np.max(np.abs(Nq['count'] - L['count'])) == 1
    assert 0.05 < np.mean(Nq['phaseF'] != L['phaseF']) < 0.6


def test_rate_zero_fallback_is_dead_code_in_the_reference(hg):
    """DB:737-740 means to fall back to 10 samples per symbol when k* = 0; numpy divides by a float32 zero with a warning,
    not an exception, so the reference really continues with spSym = inf and codeOffset = nan (both readings).  k* = 0
    cannot come out of the rate argmax (its window starts above bin 0); this repo keeps the documented intent."""
    for r in ('nep50', 'legacy'):
        assert np.isinf(hg[f'g17/k_zero/{r}/spSym']) and np.isnan(hg[f'g17/k_zero/{r}/codeOffset'])


@pytest.mark.parametrize('sname', ['gmsk', 'bpsk', 'cc11xx', 'stx'])
def test_demodulate_host_chain_equals_reference(hg, sname, monkeypatch):
    """A12 / A13 and the tail of __demodulate (DB:765-859, 863-1051) over consecutive blocks with injected rate triples,
    symbols, centres and magnitudes: bit LUT or NRZ-S decode, block-overlap alignment with planted +-1 slips (stateful),
    trust = raw bytes of the float32 magnitudes (Q3), clipped-peak tagging (STX), uint8 casts / centres mod 256 (Q4)."""
    name = str(hg[f'g18/{sname}/case'])
    d = build(hg, name, monkeypatch)
    for b in range(int(hg[f'g18/{sname}/nblocks'])):
        p = f'g18/{sname}/b{b}/'
        d.bank.pick = (3.5, 10.0)
        d.bank.X = np.ones(d.Nfft, np.complex64)
        d.bank.triple = hg[p + 'triple']
        d.bank.sym, d.bank.cen, d.bank.mag = hg[p + 'symbols'], hg[p + 'centres_dev'], hg[p + 'magnitudes']
        raw = d.get_signalBufferHostPointer()
        if sname == 'stx':
            rs = np.random.RandomState(int(hg[p + 'samples_seed']))
            x = (rs.standard_normal(d.Nfft) + 1j * rs.standard_normal(d.Nfft)).astype(np.complex64)
            for pos in rs.randint(2000, d.Nfft - 2000, 3):
                x[pos:pos + 2] *= 80
            raw[:] = x
        d.bank.calls.clear()
        with np.errstate(all='ignore'):
            est = d.uploadAndFindCarrier(raw)
            bits, cw, tw, spSym = d.demodulate()
        assert same(np.asarray(d.clippedPeakIPure, dtype=np.int64), hg[p + 'clippedPeakIPure'])
        if sname == 'stx':
            assert est[:2] == (0, 0) and est[3] == 0 and len(hg[p + 'clippedPeakIPure']) > 0
        dm = [c for c in d.bank.calls if c[0] == 'demodulate'][0]
        fc = [c for c in d.bank.calls if c[0] == 'find_centres'][0]
        assert dm[1] == int(hg[p + 'shift_arg'])                                  # the shift the matched filters run at (DB:776-781)
        assert same(fc[1], hg[p + 'legacy/spSymF']) and same(fc[2], hg[p + 'legacy/phaseF'])
        for got, key in ((bits, 'bits'), (cw, 'centres'), (tw, 'trust')):
            assert got.dtype == np.uint8 and same(got, hg[p + 'legacy/' + key]), (b, key)
        assert same(np.float64(spSym), hg[p + 'legacy/spSym'])
        assert list(hg[p + 'legacy/out_dtypes']) == ['uint8'] * 3
        if sname == 'stx':
            assert np.count_nonzero(tw == 254) >= 4                              # the -2 tags next to the clipped peaks


def test_streaming_loop_equals_reference_process_loop(hg, monkeypatch):
    """N1 against the reference's own loop (G19: ``Demodulator_process.run`` over its own SigFIFO, fed 4095 / 4096-sample chunks by
    a fake SUB socket, and its own Demodulator under the recording fake): every block is assembled from the same samples
    behind the same overlap (hash of the buffer each block is uploaded from), and the result dict handed to the decoder has
    the reference's keys and values -- block counter, estimates, bits, trust, baudrate_est, rangerate (DP:359-379), the
    constant entries, and a NaN pick as a skipped block."""
    import hashlib
    from pycusdr_amd.demodulator_process import DemodulatorRunner
    conf = json.loads(str(hg['g19/conf']))
    proto = loadProtocol('bench_GMSK')(conf=conf)
    with monkeypatch.context() as m:
        m.setattr(dbm, 'MFBank', ReplayBank)
        run = DemodulatorRunner(conf, proto, 'UHF-H')
    bank = run.demod.bank
    nblocks = int(hg['g19/nblocks'])
    rs = np.random.RandomState(int(hg['g19/stream_seed']))
    L = int(hg['g19/stream_len'])
    stream = (rs.standard_normal(L) + 1j * rs.standard_normal(L)).astype(np.complex64)
    sizes, pos, chunks = [int(v) for v in hg['g19/chunk_sizes']], 0, []
    while pos < L:
        n = sizes[len(chunks) % 2]
        chunks.append(stream[pos:pos + n])
        pos += n
    seen = []

    def on_upload(samples):
        b = len(seen)
        seen.append(hashlib.sha256(np.ascontiguousarray(samples).tobytes()).hexdigest())
        p = f'g19/b{b}/'
        bank.pick, bank.triple = hg[p + 'pick'], hg[p + 'triple']
        bank.sym, bank.cen, bank.mag = hg[p + 'symbols'], hg[p + 'centres_dev'], hg[p + 'magnitudes']
    bank.on_upload = on_upload
    bank.X = np.zeros(run.demod.Nfft, np.complex64)      # the spectrum buffer nobody wrote to in the reference run either
    with np.errstate(all='ignore'):
        results, _ = run.run_stream(chunks)
    assert len(results) == nblocks and seen == [str(v) for v in hg['g19/raw_sha']]          # same blocks, same overlap carry
    assert set(str(k) for k in hg['g19/keys']) <= set(results[0])
    for k in ('workerId', 'voteGroup', 'baudRate', 'sample_rate', 'protocol', 'rangerateEst', 'baudRate_est'):
        assert all(np.asarray(d[k]) == hg[f'g19/const/{k}'] for d in results), k
    for b, d in enumerate(results):
        p = f'g19/b{b}/'
        for k in ('count', 'doppler', 'doppler_std', 'rangerate'):
            assert same(np.float64(d[k]), hg[p + k]), (b, k, d[k], hg[p + k])
        for k in ('spSymEst', 'baudrate_est'):                                           # promotion-dependent: the legacy reading
            assert same(np.float64(d[k]), hg[p + 'legacy/' + k]), (b, k, d[k], hg[p + 'legacy/' + k])
            assert same(np.float32(d[k]), np.float32(hg[p + 'nep50/' + k])) or k == 'baudrate_est'
        assert same(np.float64(d['SNR']), hg[p + 'SNR']) or (b == 3 and d['SNR'] == 0)
        assert d['data'].dtype == np.uint8 and same(d['data'], hg[p + 'legacy/data']) and same(d['trust'], hg[p + 'legacy/trust'])
    assert np.isnan(hg['g19/b3/pick'][0]) and results[3]['doppler'] == 0          # the skipped block still goes to the decoder
    # the three-thread form goes through this repo's SigFIFO / RingBuffer instead of the in-place assembler: same blocks, same dicts
    with monkeypatch.context() as m:
        m.setattr(dbm, 'MFBank', ReplayBank)
        run2 = DemodulatorRunner(conf, proto, 'UHF-H')
    bank2, seen2 = run2.demod.bank, []

    def on_upload2(samples):
        b = len(seen2)
        seen2.append(hashlib.sha256(np.ascontiguousarray(samples).tobytes()).hexdigest())
        p = f'g19/b{b}/'
        bank2.pick, bank2.triple = hg[p + 'pick'], hg[p + 'triple']
        bank2.sym, bank2.cen, bank2.mag = hg[p + 'symbols'], hg[p + 'centres_dev'], hg[p + 'magnitudes']
    bank2.on_upload = on_upload2
    bank2.X = np.zeros(run2.demod.Nfft, np.complex64)
    with np.errstate(all='ignore'):
        piped, _ = run2.run_stream(chunks, pipelined=True)
    assert seen2 == seen and len(piped) == nblocks
    assert all(same(a['data'], b['data']) and same(np.float64(a['rangerate']), np.float64(b['rangerate'])) for a, b in zip(results, piped))
