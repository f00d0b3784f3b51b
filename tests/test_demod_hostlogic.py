"""Host logic of the Demodulator that needs no GPU: block-overlap alignment and NRZ-S extraction
against KATs recorded from the reference's pure-numpy methods (fixtures G7), plus an end-to-end run
of the host driver over the CPU oracle bank (zero bit errors on the reference's own bench packet)."""
import types

import numpy as np
import pytest

from oracle import mfbank_oracle as orc
from pycusdr_amd import config as cfg, signals as sg
from pycusdr_amd.decoder import Decoder
from pycusdr_amd.demodulator import UHF, STX
from pycusdr_amd.demodulator.demodulator_base import Demodulator
from pycusdr_amd.protocol import loadProtocol
import pycusdr_amd.demodulator.demodulator_base as dbm

from oracle_bank import OracleBank


@pytest.mark.parametrize('scenario', ['aligned', 'early', 'late', 'both', 'short'])
def test_check_symbol_overlap_kats(goldens, scenario):
    ov = int(goldens[f'g7/overlap/{scenario}/ov'])
    s = types.SimpleNamespace(sigOverlapWin=ov // 2, Nfft=1 << 12, overlapOffset=20, symbol_check_error_threshold=1000,
                              symbol_check_match_threshold=10, poswinP=[])
    for b in range(4):
        k = f'g7/overlap/{scenario}/b{b}'
        bits = goldens[f'{k}/bits']
        cw, bw, tw, _ = Demodulator.checkSymbolOverlap(s, 0, goldens[f'{k}/centres'], bits.astype(np.int32), bits,
                                                       goldens[f'{k}/trust'])
        assert np.array_equal(cw, goldens[f'{k}/centresWin'])
        assert np.array_equal(bw, goldens[f'{k}/bitsWin'])
        assert np.array_equal(tw, goldens[f'{k}/trustWin'])


def test_check_symbol_overlap_skips_when_too_many_errors(goldens):
    s = types.SimpleNamespace(sigOverlapWin=512, Nfft=1 << 12, overlapOffset=20, symbol_check_error_threshold=5,
                              symbol_check_match_threshold=10, poswinP=[])
    k = 'g7/overlap/early'
    out = []
    for b in range(4):
        bits = goldens[f'{k}/b{b}/bits']
        cw, bw, _, _ = Demodulator.checkSymbolOverlap(s, 6, goldens[f'{k}/b{b}/centres'], bits.astype(np.int32), bits,
                                                      goldens[f'{k}/b{b}/trust'])
        out.append(len(bw))
    assert out == [192] * 4          # no realignment attempted


def test_extract_bits_nrzs_kat(goldens):
    lut = loadProtocol('bench_BPSK')(conf=cfg.bench_config('bench_BPSK')).get_symbolLUT2(5)[1]
    s = types.SimpleNamespace(symbolLUT=np.asarray(lut), bitLUT=None)
    bits, err = Demodulator.extractBitsNRZs(s, None, goldens['g7/nrzs/symbols'])
    assert np.array_equal(bits, goldens['g7/nrzs/bits'])
    assert np.array_equal(np.asarray(err, dtype=np.int64), goldens['g7/nrzs/symError'])


def test_threshold_input_clips_and_marks():
    s = types.SimpleNamespace(peakThresholdScale=4.5, Nfft=4096)
    rs = np.random.RandomState(0)
    x = (rs.standard_normal(4096) + 1j * rs.standard_normal(4096)).astype(np.complex64)
    x[1000] *= 200
    x[1030] *= 300
    x[3000] *= 500
    before = x.copy()
    Demodulator._thresholdInput(s, x)
    assert set([1000, 1030, 3000]).issubset(set(s.clippedPeakIPure.tolist()))
    assert np.abs(x).max() < np.abs(before).max() / 10
    assert np.allclose(np.angle(x[3000]), np.angle(before[3000]), atol=1e-5)     # phase kept
    # peaks closer than 100 samples are joined, the far one is not
    assert set(range(1000, 1031)).issubset(set(s.clippedPeakI.tolist())) and 2999 not in s.clippedPeakI


@pytest.fixture()
def oracle_backend(monkeypatch):
    monkeypatch.setattr(dbm, 'MFBank', OracleBank)


def _run_stream(demod, dec, sig, N, ov, nblocks):
    raw = demod.get_signalBufferHostPointer()
    raw[:ov] = sig[:ov]
    got = []
    for b in range(nblocks):
        raw[ov:] = sig[ov + b * (N - ov): ov + (b + 1) * (N - ov)]
        fo, metric, clipped, snr = demod.uploadAndFindCarrier(raw)
        bits, centres, trust, spSym = demod.demodulate()
        assert bits.dtype == centres.dtype == trust.dtype == np.uint8
        pk, _, _ = dec.findFrames(bits, 0)
        got.extend(pk)
        raw[:ov] = raw[-ov:]
    return got


@pytest.mark.parametrize('mod,pname', [('GMSK', 'bench_GMSK'), ('FSK', 'bench_FSK'), ('BPSK', 'bench_BPSK')])
def test_bench_packet_decodes_without_bit_errors_on_oracle_backend(oracle_backend, mod, pname):
    """The reference's own acceptance notion (bench_modem.py:133-140): zero bit errors on the
    seed-123 packet, noiseless -- confirmed here for the oracle, then used as a gate on the GPU."""
    bs, ov = 15, 1 << 10
    N = 1 << bs
    conf = cfg.bench_config(pname, blockSize=bs, doppCarrierSteps=16)
    p = loadProtocol(pname)(conf=conf)
    demod = UHF.Demodulator(conf, p, 'UHF-H')
    dec = Decoder({}, p, correlator=orc.sync_correlate)
    sig, payload = sg.get_padded_packet(mod, 16, 153600)
    sig = np.concatenate((sig, np.zeros(N))).astype(np.complex64)
    nblocks = (len(sig) - ov) // (N - ov)
    packets = _run_stream(demod, dec, sig, N, ov, nblocks)
    assert len(packets) == 1 and packets[0].checkPacketData() == 0
    assert int(demod.doppOffsetIdx) == N // 4


def test_constructor_validation_and_stx(oracle_backend):
    conf = cfg.bench_config('bench_GMSK', blockSize=12, doppCarrierSteps=4)
    p = loadProtocol('bench_GMSK')(conf=conf)

    class WrongShape:
        name = 'wrong'
        SUM_ALL_MASKS_PYTHON = True

        def get_filter(self, N, sps, ms):
            return 8, np.zeros((8, N // 2), np.complex64)

        def get_symbolLUT2(self, ms):
            return p.get_symbolLUT2(ms)

    with pytest.raises(ValueError):
        UHF.Demodulator(conf, WrongShape(), 'UHF-H')

    class WrongType(WrongShape):
        def get_filter(self, N, sps, ms):
            return 8, np.zeros((8, N), np.complex128)

    with pytest.raises(TypeError):
        UHF.Demodulator(conf, WrongType(), 'UHF-H')
    bad = cfg.bench_config('bench_GMSK', blockSize=12, doppCarrierSteps=4)
    bad['GPU']['UHF']['CUDA']['numThreadsS'] = 3000
    with pytest.raises(ValueError):
        UHF.Demodulator(bad, p, 'UHF-H')
    with pytest.raises(KeyError):
        UHF.Demodulator(conf, p, 'no-such-radio')
    # STX: no Doppler search, fixed shift = IF offset bin
    d = STX.Demodulator(conf, p, 'UHF-H')
    raw = d.get_signalBufferHostPointer()
    raw[:] = sg.s1_stream(1, 1 << 12, 1 << 10, 'GMSK', snr_db=None)[20000:20000 + 4096] if False else \
        sg.get_padded_packet('GMSK')[0][20000:20000 + 4096].astype(np.complex64)
    assert d.uploadAndFindCarrier(raw)[:2] == (0, 0)
    bits, cen, trust, spSym = d.demodulate()
    assert int(d.dopplerIdxlast) == (1 << 12) // 4 and abs(spSym - 16) < 0.1 and len(bits) > 150


def _cc11xx_stimulus(bs, sps, payload, snr_db=15.0):
    from pycusdr_amd.protocol.CC11xx import frame_bits
    bits = frame_bits(payload, preamble=(0xAA,) * 10)
    fs = 7416 * sps
    sig = np.concatenate((np.zeros(9000), sg.modulateFSK(bits, sps), np.zeros(7 * (1 << bs))))
    sig = sig * np.exp(1j * 2 * np.pi * 148320 / fs * np.arange(len(sig)))
    return sg.awgn(sig, snr_db, rng=np.random.RandomState(2)).astype(np.complex64)


def test_cc11xx_frame_received_and_crc_ok_on_oracle_backend(oracle_backend):
    """BASELINE config C1 (CC11xx protocol, 32 Doppler bins, 2^16-sample chunks, CPU path -- plumbing, no GPU):
    CC11xx FSK-2 at 128 samples/symbol, IF offset 148.32 kHz (config/CC11xx.json geometry); a framed, whitened,
    CRC-protected packet goes through Doppler search, demodulation, sync correlation and the packet parser."""
    from pycusdr_amd.demodulator_process import DemodulatorRunner
    bs, sps = 16, 128
    conf = cfg.cc11xx_config(blockSize=bs, doppCarrierSteps=32, samplesPerSym=sps)
    p = loadProtocol('CC11xx')(conf=conf)
    p.CRC_CHECK = 'framer'        # the stimulus is a TX-framer frame: CRC inside the length-counted bytes
    run = DemodulatorRunner(conf, p, 'UHF-H')
    payload = np.arange(1, 11, dtype=np.uint8)
    sig = _cc11xx_stimulus(bs, sps, payload)
    step = (1 << bs) - (1 << 10)
    # FIXED mode waits until packetLen = 2136 bits follow the header: ~504 bits arrive per block
    chunks = [sig[i * step:(i + 1) * step] for i in range(6)]
    res, packets = run.run(chunks, decoder=Decoder({}, p, correlator=orc.sync_correlate))
    assert abs(res[0]['doppler']) < 600 and abs(res[0]['spSymEst'] - 128) < 1
    assert len(packets) == 1
    data, crc_err, _ = packets[0].getBinaryData()
    assert packets[0].packetLen == 12 and not crc_err and np.array_equal(data[:-2], payload)


def test_streaming_runner_on_oracle_backend(oracle_backend):
    """The caller's loop (N1) on the CPU: overlap carry, result-dict keys, rate bookkeeping."""
    from pycusdr_amd.demodulator_process import DemodulatorRunner, radioBackendVoteGroupIDX
    bs, ov = 14, 1 << 10
    N = 1 << bs
    conf = cfg.bench_config('bench_FSK', blockSize=bs, doppCarrierSteps=8)
    p = loadProtocol('bench_FSK')(conf=conf)
    run = DemodulatorRunner(conf, p, 'UHF-H')
    sig = sg.get_padded_packet('FSK')[0][9000:9000 + 3 * N].astype(np.complex64)
    step = N - ov
    chunks = [sig[i * step:(i + 1) * step] for i in range(3)]
    seen = []
    run.run(chunks, sink=seen.append)
    assert [d['count'] for d in seen] == [0, 1, 2] and run.iterCount == 3 and run.timeMA > 0
    # the overlap carry: the head of the pinned buffer holds the tail of the previous block
    assert np.array_equal(run.raw[:ov], np.asarray(chunks[2][-ov:]))
    d = seen[1]
    assert d['voteGroup'] == 0 and d['protocol'] == 'UHF' and d['sample_rate'] == 153600 and abs(d['spSymEst'] - 16) < 0.1
    assert abs(d['doppler']) < 50 and d['rate_ksps'] > 0
    # every key of the reference's result dict (DP:259-276 + the two its loop adds), and it survives the pickle
    # round trip of demodOut.send_pyobj (DP:309)
    import pickle
    for key in ('workerId', 'count', 'timestamp', 'voteGroup', 'doppler', 'doppler_std', 'data', 'trust', 'spSymEst', 'SNR',
                'rangerateEst', 'baudRate', 'baudRate_est', 'sample_rate', 'protocol', 'baudrate_est', 'rangerate'):
        assert key in d, key
    back = pickle.loads(pickle.dumps(d, protocol=pickle.HIGHEST_PROTOCOL))
    assert set(back) == set(d) and np.array_equal(back['data'], d['data']) and back['data'].dtype == np.uint8
    assert back['doppler'] == d['doppler'] and back['workerId'] == d['workerId']
    # the same stream in chunks of 2^14 samples (the BER bench's size) through the ring buffer: same blocks
    run2 = DemodulatorRunner(conf, p, 'UHF-H')
    seen2 = []
    run2.run_stream((sig[i:i + (1 << 14)] for i in range(0, 3 * step, 1 << 14)), sink=seen2.append)
    assert len(seen2) == 3 and all(np.array_equal(a['data'], b['data']) and a['doppler'] == b['doppler']
                                   for a, b in zip(seen, seen2))
    assert [radioBackendVoteGroupIDX(b)[1] for b in ('UHF', 'STX', 'STX1', 'STX2')] == [0, 1, 2, 3]      # reference DP:20-36
    with pytest.raises(Exception, match='not defined in voteGroup'):
        radioBackendVoteGroupIDX('LBAND')
