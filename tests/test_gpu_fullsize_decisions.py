"""End-to-end symbol-decision parity at BASELINE size (N = 2^20): the HIP Demodulator against the same host driver over the CPU
oracle bank, for every single-GPU BASELINE filter bank -- C2's GMSK bank (256 bins, 10 dB) and C5's two banks (CC11xx FSK-2 at
128 samples per symbol and the 32-filter BPSK bank, 512 bins each) on their matching stimuli -- over two consecutive blocks, so
that the block-overlap alignment (DB:863-988) runs at this size too.  north_star: "bit-exact on symbol decisions".

The oracle's Doppler search is restricted to the nine bins around the carrier: the pick's top-2 scan (CU:502-597) looks at
positive scores only and the two largest sit there, so the index, the interpolated shift and everything behind it are those of
the full table; what the restriction saves is 247 (503) x M inverse 2^20-point transforms per block on the CPU.  The scores
of those nine bins are held to north_star's 1e-5 against the HIP table.  Reference: CU:78-146, 174-320, DB:604-632, 711-859."""
import numpy as np
import pytest

from oracle import mfbank_oracle as orc
from pycusdr_amd import config as cfg, signals as sg
from pycusdr_amd.demodulator import UHF
from pycusdr_amd.protocol import loadProtocol
import pycusdr_amd.demodulator.demodulator_base as dbm

from oracle_bank import OracleBank

pytestmark = pytest.mark.gpu

# |arg P[k*]| difference between the fp32 device chain and the fp64 oracle chain at N = 2^20, measured on the three banks below
# (printed by this test): 1.4e-5 rad (GMSK), 1.6e-8 (CC11xx), 4.6e-7 (BPSK).  A symbol's window start moves by spSym / (2 pi) samples per radian, i.e. 1e-4 samples at
# 16 samples per symbol -- the truncation (int) of CU:88 can only flip for a block whose code offset sits within that of an
# integer, and then every window of the block moves by one sample TOGETHER (decisions unchanged: the window is 7 samples wide).
PHASE_TOL = 5e-5


class NearCarrierOracleBank(OracleBank):
    """OracleBank whose search transforms only the bins within `RADIUS` of the bin the carrier is known to be in."""
    CENTRE, RADIUS = None, 4

    def _scores(self):
        ds = np.zeros((self.Dtot, self.M), dtype=np.float32)
        lo, hi = max(self.Doff, self.CENTRE - self.RADIUS), min(self.Dtot, self.CENTRE + self.RADIUS + 1)
        ds[lo:hi] = orc.doppler_scores(self.X, self.masks, self.shifts[lo:hi], self.sum_all).astype(np.float32)
        self.window = (lo, hi)
        return ds

    def find_carrier(self):
        self.ds = self._scores()
        return orc.find_doppler_est(self.ds, self.D, self.Doff, self.sum_all)


def _stimulus(kind, N, ov, nblocks):
    need = nblocks * (N - ov) + ov
    if kind in ('GMSK', 'BPSK'):
        return sg.s1_stream(nblocks, N, ov, kind, snr_db=10.0, seed=1)
    # CC11xx: framed FSK-2 at 128 samples per symbol on the 148.32 kHz IF offset of config/CC11xx.json
    from pycusdr_amd.protocol.CC11xx import frame_bits
    rs = np.random.RandomState(4)
    sps, fs = 128, 7416 * 128
    bits = np.concatenate([frame_bits(rs.randint(0, 256, 60).astype(np.uint8), preamble=(0xAA,) * 10) for _ in range(6)])
    base = sg.modulateFSK(bits, sps)
    sig = np.tile(base, -(-need // len(base)))[:need]
    sig = sig * np.exp(2j * np.pi * 148320 / fs * np.arange(need))
    return sg.awgn(sig, 12.0, rng=np.random.RandomState(2)).astype(np.complex64)


# basis 'span' (round 6): the opt-in span basis of the SUM_ALL search (mfb_set_search_basis: rank(bank) filters transformed instead
# of M -- GMSK 6 of 8, CC11xx 4 of 8, BPSK 5 of 32) held to the SAME gate: the oracle's full bank, the same pick, the same bits.
@pytest.mark.parametrize('basis', ['filters', 'span'])
@pytest.mark.parametrize('kind,pname,D', [('GMSK', 'bench_GMSK', 256), ('CC11xx', 'CC11xx', 512), ('BPSK', 'bench_BPSK', 512)])
def test_fullsize_symbol_decisions_equal_the_oracle_chain(monkeypatch, kind, pname, D, basis, capsys):
    bs, ov = 20, 1 << 10
    N = 1 << bs
    if pname == 'CC11xx':
        conf = cfg.cc11xx_config(blockSize=bs, doppCarrierSteps=D, samplesPerSym=128)
    else:
        conf = cfg.bench_config(pname, blockSize=bs, doppCarrierSteps=D)
    p = loadProtocol(pname)(conf=conf)
    gpu = UHF.Demodulator(conf, p, 'UHF-H')
    with monkeypatch.context() as m:
        m.setattr(dbm, 'MFBank', NearCarrierOracleBank)
        cpu = UHF.Demodulator(conf, loadProtocol(pname)(conf=conf), 'UHF-H')
    assert type(gpu.bank).__name__ == 'MFBank' and isinstance(cpu.bank, NearCarrierOracleBank)
    if basis == 'span':
        gpu.bank.set_search_basis('span')
        name, transformed = gpu.bank.get_search_basis()
        assert name == 'span' and transformed < gpu.bank.M, (name, transformed)      # the shortcut really is in force
    for dm in (gpu, cpu):                      # tap the un-truncated centres of the kept symbols
        def tapped(*a, _orig=dm.checkSymbolOverlap, _dm=dm, **k):
            out = _orig(*a, **k)
            _dm._centresWin = np.asarray(out[0])
            return out
        dm.checkSymbolOverlap = tapped
    sig = _stimulus(kind, N, ov, 2)
    worst_phase, moved = 0.0, 0
    try:
        for b in range(2):
            x = sig[b * (N - ov): b * (N - ov) + N]
            raw = gpu.get_signalBufferHostPointer()
            raw[:] = x
            og = gpu.uploadAndFindCarrier(raw)
            # the carrier's bin from the HIP pick; the oracle transforms the nine bins around it
            cpu.bank.CENTRE = cpu.doppIdxArrayOffset + int(gpu._pick_bin)
            oc = cpu.uploadAndFindCarrier(x.copy())
            lo, hi = cpu.bank.window
            sg_, sc_ = gpu.bank.get_scores()[lo:hi], cpu.bank.get_scores()[lo:hi]
            assert np.abs(sg_ - sc_).max() / sc_.max() < 1e-5                 # correlation magnitudes, north_star's 1e-5
            assert gpu.bank.get_scores()[:, 0].argmax() in range(lo, hi)        # the restriction did not hide the maximum
            assert int(gpu.dopplerIdxlast) == int(cpu.dopplerIdxlast), b
            assert abs(og[0] - oc[0]) <= 1e-3 * max(1.0, abs(oc[0])) + 0.05     # Hz
            bg, cg, tg, spg = gpu.demodulate()
            bc, cc, tc, spc = cpu.demodulate()
            assert spg == spc, b                                                # same rate bin -> the same float64
            dphi = abs(float(gpu._codeRateResult[1]) - float(cpu._codeRateResult[1]))
            worst_phase = max(worst_phase, min(dphi, 2 * np.pi - dphi))
            assert len(bg) == len(bc) > 0.9 * N / gpu.spsym
            assert np.array_equal(bg, bc), f'block {b}: {np.count_nonzero(bg != bc)} symbol decisions differ'
            dcen = gpu._centresWin - cpu._centresWin
            moved = max(moved, int(np.count_nonzero(dcen)))
            # a peak that ties between neighbouring samples to within fp32-vs-fp64 round-off may sit one sample off
            assert np.abs(dcen).max() <= 1 and np.count_nonzero(dcen) <= max(2, len(dcen) // 2000), (b, np.count_nonzero(dcen))
            assert np.array_equal(np.asarray(gpu.poswinP), np.asarray(cpu.poswinP))
        assert worst_phase < PHASE_TOL, worst_phase
        with capsys.disabled():
            print(f'\n[fullsize decisions] {pname} D={D} basis={basis}: |arg| difference <= {worst_phase:.2e} rad, centres off by one: {moved}')
    finally:
        gpu.close()
