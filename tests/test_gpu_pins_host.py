"""The one-call block path's DEVICE-side float64 arithmetic against reference-run fixtures (G16-G18,
tests/golden/ref_goldens_host.npz).

``mfb_receive_block`` moved the reference's host arithmetic behind the pick and behind the rate argmax onto the device
(``block_pick_body`` inside k_pick_block, ``block_rate_body`` inside k_code_rate_block).  ``mfb_debug_block_scalars`` runs
exactly those device functions on injected {index, metric} / {k*, arg, |P|^2} values -- the point at which the reference's
own code was fed the same values when the fixtures were recorded -- and this repo's host half of the one-call path
(``_estimate_from_block``, ``demodulateDevice`` / ``demodulateHost``) finishes the job.  Everything must equal the reference's
result (``legacy`` reading) bit for bit."""
import json
import os

import numpy as np
import pytest

from pycusdr_amd.demodulator import STX, UHF
from pycusdr_amd.protocol import loadProtocol

from test_pins_host import same, spectrum

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def hg():
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'ref_goldens_host.npz'), allow_pickle=False)
    return {k.replace('__', '/'): z[k] for k in z.files}


def build(hg, name):
    conf = json.loads(str(hg[f'g15/{name}/conf']))
    backend = {'UHF': UHF, 'STX': STX}[str(hg[f'g15/{name}/backend'])]
    d = backend.Demodulator(conf, loadProtocol(str(hg[f'g15/{name}/protocol']))(conf=conf), 'UHF-H')
    assert type(d.bank).__name__ == 'MFBank' and d._one_call
    return d


def windows(X, pieces):
    """The two spectrum windows from the [signal | noise][piece][start, length] bounds the device computed."""
    return tuple(np.concatenate([X[s:s + n] for s, n in band]) for band in pieces)


# (the fixture case cc11xx_b16_s16 -- an IF offset beyond the sample rate, shifts outside [0, N) -- is host-only:
# mfb_set_shifts refuses such a table, where the reference's kernel would read out of bounds)
@pytest.mark.parametrize('name', ['bench_b15_d64', 'zero_if', 'neg_if', 'noise_neg', 'noise_pos', 'cc11xx_b17_s128', 'bench_b20_d256'])
def test_pick_stage_of_the_block_path_equals_reference(hg, name):
    """DB:604-667 with the shift interpolation and the window bounds computed ON THE DEVICE: frequency offset, SNR (same
    elements in the same order as the reference's numpy slices, wrapped bands included), dopplerIdxlast, NaN => block skipped."""
    d = build(hg, name)
    p = f'g16/{name}/'
    X = spectrum(d.Nfft, int(hg[p + 'spectrum_seed']))
    picks = np.stack((hg[p + 'pick'], hg[p + 'metric']), axis=1)
    triples = np.tile(np.float32([d.Nfft / d.spsym, 0.5, 1.0]), (len(picks), 1))
    out = d.bank.debug_block_scalars(picks, triples, d.spsymMin, 5)
    # the capacity rule covers every pick the device can produce (an index in front of the first Doppler bin -- between the
    # noise-reference row and the table -- cannot come out of findDopplerEst; such injected values exercise the arithmetic only)
    doff = d.doppIdxArrayOffset
    assert d.bank.BAND_CAPACITY >= max(max(r['band_len']) for r, pk in zip(out, picks) if not pk[0] < doff)
    for i, r in enumerate(out):
        blk = dict(r, bands=windows(X, r['band_pieces']))
        assert [len(b) for b in blk['bands']] == list(r['band_len'])
        with np.errstate(all='ignore'):
            fo, sd, clipped, snr = d._estimate_from_block(blk)
        assert same(np.float64(fo), hg[p + 'freqOffset'][i]), (i, picks[i])
        assert same(np.float64(snr), hg[p + 'SNR'][i]), (i, picks[i])
        assert int(d.dopplerIdxlast) == int(hg[p + 'dopplerIdxlast'][i]) == r['shift'] or not r['pick_valid']
        assert same(np.float32(sd), np.float32(hg[p + 'sdev_Hz_nep50'][i]))
    assert not out[-1]['pick_valid'] and int(d.dopplerIdxlast) == 0
    d.close()


@pytest.mark.parametrize('name', ['bench_b15_d64', 'bench_b20_d256', 'cc11xx_b17_s128'])
def test_rate_stage_of_the_block_path_equals_reference(hg, name):
    """DB:733-752 + 994-999 on the device, over every k* of the rate window x eleven phases (and k* far outside it: the
    clamp): spSym, codeOffset, the two float32 launch arguments of findCentres and the symbol count."""
    d = build(hg, name)
    p = f'g17/{name}/'
    ks, args = hg[p + 'k'], hg[p + 'arg']
    triples = np.stack((ks, args, np.ones_like(ks)), axis=1)
    picks = np.tile(np.float32([1.0, 0.0]), (len(ks), 1))
    out = d.bank.debug_block_scalars(picks, triples, d.spsymMin, 5, max_symbols=d.Nfft)
    L = {f: hg[p + 'legacy/' + f] for f in ('spSym', 'codeOffset', 'spSymF', 'phaseF', 'count')}
    cap = d.Nfft // 2          # the handle's symbol capacity
    assert same(np.array([r['spSym'] for r in out]), L['spSym'])
    assert same(np.array([r['codeOffset'] for r in out]), L['codeOffset'])
    assert same(np.array([r['spSymF'] for r in out], np.float32), L['spSymF'])
    assert same(np.array([r['offsetF'] for r in out], np.float32), L['phaseF'])
    assert same(np.array([r['count'] for r in out]), np.minimum(L['count'], cap)) and L['count'].max() <= cap
    assert not any(r['rate_fallback'] for r in out)
    d.close()


@pytest.mark.parametrize('sname', ['gmsk', 'bpsk', 'cc11xx', 'stx'])
def test_block_path_host_half_equals_reference(hg, sname):
    """The whole demodulation tail on the one-call path -- device scalars, then ``demodulateDevice`` / ``demodulateHost`` on the
    injected symbol arrays -- against what the reference's ``demodulate`` returned for the same injected device results."""
    d = build(hg, str(hg[f'g18/{sname}/case']))
    for b in range(int(hg[f'g18/{sname}/nblocks'])):
        p = f'g18/{sname}/b{b}/'
        r = d.bank.debug_block_scalars([[3.5, 10.0]], [hg[p + 'triple']], d.spsymMin, 5)[0]
        assert same(r['spSymF'], hg[p + 'legacy/spSymF']) and same(r['offsetF'], hg[p + 'legacy/phaseF'])
        n = r['count']
        blk = dict(r, symbols=hg[p + 'symbols'][:n].copy(), centres=hg[p + 'centres_dev'][:n].copy(),
                   magnitudes=hg[p + 'magnitudes'][:n].copy(), bands=None)
        if sname == 'stx':
            rs = np.random.RandomState(int(hg[p + 'samples_seed']))
            x = (rs.standard_normal(d.Nfft) + 1j * rs.standard_normal(d.Nfft)).astype(np.complex64)
            for pos in rs.randint(2000, d.Nfft - 2000, 3):
                x[pos:pos + 2] *= 80
            d._thresholdInput(x)
            d.dopplerIdxlast = d.doppOffsetIdx
            blk['shift'] = int(d.doppOffsetIdx)
        else:
            d.dopplerIdxlast = np.int32(r['shift'])
        assert int(d.dopplerIdxlast) == int(hg[p + 'shift_arg'])
        d._pending = blk
        bits, cw, tw, spSym = d.demodulate()
        assert d._pending is None                                                  # the one-call branch consumed the record
        for got, key in ((bits, 'bits'), (cw, 'centres'), (tw, 'trust')):
            assert got.dtype == np.uint8 and same(got, hg[p + 'legacy/' + key]), (b, key)
        assert same(np.float64(spSym), hg[p + 'legacy/spSym'])
    d.close()
