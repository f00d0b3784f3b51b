"""GPU parity tests of the opt-in spectral-energy search (mfb_set_search_mode, energy_kernels.hpp): the doppSum
table from Parseval's identity must equal the one the matched-filter bank produces (oracle: the reference's
unnormalised inverse transforms and row sums, cuda_kernels.cu:421-480) within 1e-5, picks identical, and the
demodulation stage must not notice."""
import numpy as np
import pytest

from oracle import mfbank_oracle as orc
from pycusdr_amd import config as cfg, signals as sg
from pycusdr_amd.demodulator import UHF
from pycusdr_amd.mfbank import MFBank
from pycusdr_amd.protocol import loadProtocol

pytestmark = pytest.mark.gpu


def _rc(rs, *s):
    return (rs.standard_normal(s) + 1j * rs.standard_normal(s)).astype(np.complex64)


@pytest.mark.parametrize('sum_all', [True, False])
@pytest.mark.parametrize('kind', ['gmsk', 'bpsk', 'dense'])
@pytest.mark.parametrize('doff', [0, 1])
def test_energy_scores_match_oracle(kind, sum_all, doff):
    log2N, D = 14, 19
    N = 1 << log2N
    rs = np.random.RandomState(5)
    if kind == 'dense':                          # no short impulse response: the demodulation stage is two-pass
        M, masks = 5, _rc(rs, 5, N)
    else:
        name, ms = ('bench_GMSK', 3) if kind == 'gmsk' else ('bench_BPSK', 5)
        M, masks = loadProtocol(name)(conf=cfg.bench_config(name, blockSize=log2N)).get_filter(N, 16, ms)
    x = _rc(rs, N)
    shifts = rs.randint(0, N, D + doff).astype(np.int32)
    shifts[0], shifts[-1] = 0, N - 1
    bank = MFBank(log2N, D, M, sum_all_masks=sum_all, doppler_offset=doff)
    try:
        bank.set_filters(masks)
        bank.set_shifts(shifts)
        bank.upload(x)
        assert bank.get_search_mode() == 'transforms'                # the shortcut is never on by itself
        idx_t, met_t = bank.find_carrier()
        ds_t = bank.get_scores()
        bank.set_search_mode('energy')
        assert bank.get_search_mode() == 'energy'
        idx_e, met_e = bank.find_carrier()
        ds_e = bank.get_scores()
        idx_e2, _ = bank.find_carrier()
        assert np.array_equal(ds_e, bank.get_scores()) and idx_e == idx_e2      # bit-reproducible
        ref = orc.doppler_scores(bank.get_spectrum(), masks, shifts, sum_all)
        assert np.abs(ds_e - ref).max() / ref.max() < 1e-5
        assert np.abs(ds_e - ds_t).max() / ds_t.max() < 1e-5
        if sum_all:
            assert not ds_e[:, 1:].any()                                # only column 0 is populated (CU:453-464)
        oidx, ometric = orc.find_doppler_est(ds_e, D, doff, sum_all)
        assert idx_e == oidx and abs(met_e - ometric) < 1e-3
        assert abs(idx_e - idx_t) < 1e-3
        # new filters invalidate the cached energy spectrum
        bank.set_filters((2.0 * masks).astype(np.complex64))
        bank.find_carrier()
        assert np.abs(bank.get_scores() - 4.0 * ref).max() / (4.0 * ref.max()) < 1e-5
        # the matched filtering at the chosen shift is the bank's own, whatever the search did
        bank.set_filters(masks)
        bank.find_carrier()
        bank.demodulate(int(shifts[3]), 5, 50)
        rxc = orc.demod_xcorr(bank.get_spectrum(), masks, int(shifts[3]))
        assert np.abs(bank.get_xcorr() - rxc).max() / np.abs(rxc).max() < 1e-5
        bank.set_search_mode('transforms')
        bank.find_carrier()
        assert np.array_equal(bank.get_scores(), ds_t)
    finally:
        bank.close()


def test_energy_mode_rejects_unknown_and_survives_path_changes():
    log2N, D = 13, 8
    N = 1 << log2N
    rs = np.random.RandomState(1)
    M, masks = loadProtocol('bench_FSK')(conf=cfg.bench_config('bench_FSK', blockSize=log2N)).get_filter(N, 16, 3)
    bank = MFBank(log2N, D, M, sum_all_masks=True)
    try:
        with pytest.raises(KeyError):
            bank.set_search_mode('fast')
        bank.set_filters(masks)
        bank.set_shifts(rs.randint(0, N, D).astype(np.int32))
        bank.upload(_rc(rs, N))
        bank.set_search_mode('energy')
        a = bank.find_carrier()
        sa = bank.get_scores()
        for path in ('twopass', 'segment', 'auto'):
            bank.set_search_path(path)
            assert bank.get_search_mode() == 'energy'
            assert bank.find_carrier() == a and np.array_equal(bank.get_scores(), sa)
    finally:
        bank.close()


def test_demodulator_output_identical_with_energy_search():
    """Whole block through the Demodulator with the search in either mode: same pick, same bits."""
    bs, ov = 15, 1 << 10
    N = 1 << bs
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=32)
    sig = sg.s1_stream(3, N, ov, 'GMSK', snr_db=15.0, seed=3)
    outs = []
    for mode in ('transforms', 'energy'):
        demod = UHF.Demodulator(conf, loadProtocol('bench_GMSK')(conf=conf), 'UHF-H')
        try:
            demod.bank.set_search_mode(mode)
            raw = demod.get_signalBufferHostPointer()
            blocks = []
            for b in range(2):
                raw[:] = sig[b * (N - ov):b * (N - ov) + N]
                fo, metric, _, snr = demod.uploadAndFindCarrier(raw)
                bits, centres, trust, spSym = demod.demodulate()
                blocks.append((fo, int(demod.dopplerIdxlast), bits.copy(), centres.copy(), trust.copy(), spSym))
            outs.append(blocks)
        finally:
            demod.close()
    for a, b in zip(*outs):
        # (the two searches agree to fp32 rounding, so the interpolated frequency agrees to a fraction of a hertz; the
        # integer shift, the symbol rate and every decision are identical)
        assert abs(a[0] - b[0]) < 0.05 and a[1] == b[1] and a[5] == b[5]
        for u, v in zip(a[2:5], b[2:5]):
            assert np.array_equal(u, v)


def test_search_settings_from_the_config():
    """GPU.<set>.HIP carries the optional search settings; what they select is in force after construction."""
    conf = cfg.bench_config('bench_GMSK', blockSize=14, doppCarrierSteps=16)
    conf['GPU']['UHF']['HIP'] = {'search_path': 'segment', 'search_basis': 'span', 'search_mode': 'energy'}
    demod = UHF.Demodulator(conf, loadProtocol('bench_GMSK')(conf=conf), 'UHF-H')
    try:
        assert demod.bank.get_search_path()['path'] == 'segment'
        assert demod.bank.get_search_basis()[0] == 'span' and demod.bank.get_search_mode() == 'energy'
    finally:
        demod.close()
    conf['GPU']['UHF']['HIP'] = {'search_mode': 'fast'}
    with pytest.raises(KeyError):
        UHF.Demodulator(conf, loadProtocol('bench_GMSK')(conf=conf), 'UHF-H')
