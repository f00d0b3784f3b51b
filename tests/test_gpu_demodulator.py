"""End-to-end parity on the GPU: the Demodulator over libmfbank.so against the same host driver
over the CPU oracle bank, on the reference's own bench packets (stimulus pinned by fixture G5)."""
import numpy as np
import pytest

from oracle import mfbank_oracle as orc
from pycusdr_amd import config as cfg, signals as sg
from pycusdr_amd.decoder import Decoder
from pycusdr_amd.demodulator import UHF
from pycusdr_amd.protocol import loadProtocol
import pycusdr_amd.demodulator.demodulator_base as dbm

from oracle_bank import OracleBank

pytestmark = pytest.mark.gpu


def _pair(pname, bs, D, monkeypatch):
    conf = cfg.bench_config(pname, blockSize=bs, doppCarrierSteps=D)
    p = loadProtocol(pname)(conf=conf)
    gpu = UHF.Demodulator(conf, p, 'UHF-H')
    with monkeypatch.context() as m:
        m.setattr(dbm, 'MFBank', OracleBank)
        cpu = UHF.Demodulator(conf, loadProtocol(pname)(conf=conf), 'UHF-H')
    assert type(gpu.bank).__name__ == 'MFBank' and type(cpu.bank).__name__ == 'OracleBank'
    return conf, p, gpu, cpu


# 60 dB stands for "noiseless": with exactly-zero padding the matched-filter outputs there are pure FFT
# round-off and the per-symbol argmax is arbitrary in both implementations; a whisper of noise makes
# every decision well defined without disturbing the packet.
@pytest.mark.parametrize('mod,pname,snr', [('GMSK', 'bench_GMSK', 60.0), ('GMSK', 'bench_GMSK', 10.0), ('GMSK', 'bench_GMSK', 6.0),
                                           ('FSK', 'bench_FSK', 60.0), ('FSK', 'bench_FSK', 12.0), ('GFSK', 'bench_GFSK', 12.0),
                                           ('BPSK', 'bench_BPSK', 60.0), ('BPSK', 'bench_BPSK', 12.0)])
def test_stream_bits_identical_to_oracle_and_packet_error_free(monkeypatch, mod, pname, snr):
    bs, ov, D = 15, 1 << 10, 32
    N = 1 << bs
    conf, p, gpu, cpu = _pair(pname, bs, D, monkeypatch)
    sig, payload = sg.get_padded_packet(mod, 16, 153600)
    sig = np.concatenate((sig, np.zeros(N)))
    if snr is not None:
        sig = sg.awgn(sig, snr, rng=np.random.RandomState(1))
    sig = sig.astype(np.complex64)
    nblocks = (len(sig) - ov) // (N - ov)
    dec = Decoder({}, p)                       # HIP sync correlator
    packets = []
    for dm in (gpu, cpu):                      # tap the un-truncated centres of the kept symbols
        def tapped(*a, _orig=dm.checkSymbolOverlap, _dm=dm, **k):
            out = _orig(*a, **k)
            _dm._centresWin = np.asarray(out[0])
            return out
        dm.checkSymbolOverlap = tapped
    rg, rc = gpu.get_signalBufferHostPointer(), cpu.get_signalBufferHostPointer()
    rg[:ov] = rc[:ov] = sig[:ov]
    for b in range(nblocks):
        rg[ov:] = rc[ov:] = sig[ov + b * (N - ov): ov + (b + 1) * (N - ov)]
        og = gpu.uploadAndFindCarrier(rg)
        oc = cpu.uploadAndFindCarrier(rc)
        sg_, sc_ = gpu.bank.get_scores(), cpu.bank.get_scores()
        if sc_.max() > 0:
            assert np.abs(sg_ - sc_).max() / sc_.max() < 1e-5          # correlation magnitudes, 1e-5
        assert int(gpu.dopplerIdxlast) == int(cpu.dopplerIdxlast)
        assert abs(og[0] - oc[0]) <= 1e-3 * max(1.0, abs(oc[0])) + 0.05  # Hz
        if np.isfinite(oc[3]) and oc[3] != 0:
            assert abs(og[3] - oc[3]) < 1e-2                              # SNR dB
        bg, cg, tg, spg = gpu.demodulate()
        bc, cc, tc, spc = cpu.demodulate()
        assert spg == spc                                                 # same FFT bin -> same float
        if snr <= 20:
            # noise floor far above fp32 round-off: symbol decisions are BIT-EXACT (north_star), fp32 device
            # transforms against the fp64 oracle (measured with tests/tools/decision_slack.py: 0 of ~12 000 symbols
            # differ for every modulation at 6-12 dB).
            assert np.array_equal(bg, bc), f'block {b}: symbol decisions differ'
            # peak sample index: equal except where two neighbouring |xc|^2 samples tie to within fp32-vs-fp64
            # round-off -- measured: at most one centre per packet, by one sample (the kernel itself is
            # bit-exact against the oracle on its own matched-filter outputs: tests/test_gpu_kernels.py)
            dcen = gpu._centresWin - cpu._centresWin
            assert np.abs(dcen).max() <= 1 and np.count_nonzero(dcen) <= 2
        else:
            # quiet padding: the matched-filter outputs there are fp32 FFT round-off of the strong
            # packet, so the argmax is arbitrary in BOTH implementations; decisions must agree on
            # every symbol that carries signal (|x|^2 around its centre above 1/4 of the carrier)
            pw = np.convolve(np.abs(rg) ** 2, np.ones(16) / 16, mode='same')

            def strong(bits, demod):
                c = demod._centresWin            # full int32 centres of the kept symbols
                keep = pw[c] > 0.25
                return c[keep], bits[keep]
            (cgs, bgs), (ccs, bcs) = strong(bg, gpu), strong(bc, cpu)
            assert np.array_equal(bgs, bcs), f'block {b}: decisions on signal-bearing symbols differ'
            # the sample index of the peak may move inside the W=7 window where samples of a flat
            # noiseless matched-filter plateau (constant-envelope FSK runs) tie to within round-off
            assert np.abs(cgs - ccs).max() <= 6
            assert np.count_nonzero(cgs != ccs) <= 0.02 * len(cgs)
        pk, _, _ = dec.findFrames(bg, 0)
        packets.extend(pk)
        rg[:ov] = rg[-ov:]
        rc[:ov] = rc[-ov:]
    assert len(packets) == 1
    assert packets[0].checkPacketData() == 0
    gpu.close()


def test_known_carrier_bin_and_symbol_rate(monkeypatch):
    bs, D = 16, 64
    N = 1 << bs
    conf, p, gpu, _ = _pair('bench_GMSK', bs, D, monkeypatch)
    sig = sg.s1_stream(2, N, 1 << 10, 'GMSK', snr_db=10.0, seed=1)
    raw = gpu.get_signalBufferHostPointer()
    raw[:] = sig[:N]
    fo, metric, clipped, snr = gpu.uploadAndFindCarrier(raw)
    assert int(gpu.dopplerIdxlast) == N // 4          # carrier at fs/4 (create_signals.py:181-182)
    assert abs(fo) < 100.0 and len(clipped) == 0
    bits, cen, trust, spSym = gpu.demodulate()
    assert abs(spSym - 16.0) < 0.5
    # trust bytes are the raw bytes of the leading fp32 magnitudes (reference quirk Q3)
    assert trust.dtype == np.uint8
    gpu.close()


def test_all_zero_block_is_skipped(monkeypatch):
    conf, p, gpu, _ = _pair('bench_GMSK', 12, 8, monkeypatch)
    raw = gpu.get_signalBufferHostPointer()
    raw[:] = 0
    assert gpu.uploadAndFindCarrier(raw) == (0., 0., [], 0.)
    assert gpu.dopplerIdxlast == 0
    gpu.close()


def test_streaming_runner_result_dict_and_ber(monkeypatch):
    """The caller's loop (next-scope row N1): overlap carry, result dict keys, decoder hand-off."""
    from pycusdr_amd.demodulator_process import DemodulatorRunner
    bs, ov = 15, 1 << 10
    N = 1 << bs
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=32)
    p = loadProtocol('bench_GMSK')(conf=conf)
    run = DemodulatorRunner(conf, p, 'UHF-H')
    sig = sg.awgn(np.concatenate((sg.get_padded_packet('GMSK')[0], np.zeros(N))), 12.0, rng=np.random.RandomState(4))
    sig = sig.astype(np.complex64)
    step = N - ov
    chunks = [sig[i * step:(i + 1) * step] for i in range(len(sig) // step)]
    results, packets = run.run(chunks, decoder=Decoder({}, p))
    assert len(results) == len(chunks) and [r['count'] for r in results] == list(range(len(chunks)))
    for key in ('workerId', 'timestamp', 'voteGroup', 'doppler', 'doppler_std', 'data', 'trust', 'spSymEst', 'SNR',
                'baudRate', 'sample_rate', 'protocol', 'rangerate', 'baudrate_est'):
        assert key in results[0]
    assert results[0]['workerId'] == 'bench_GMSK-UHF-H' and results[0]['data'].dtype == np.uint8
    assert len(packets) == 1 and packets[0].checkPacketData() == 0
    with pytest.raises(ValueError):
        run.feed(np.zeros(10, np.complex64))
    run.close()


def test_stx_backend_and_peak_clipping(monkeypatch):
    """STX back end (next-scope row N3): input peak clipping on the host, fixed shift, no Doppler
    search; clipped positions tag the trust of nearby symbols with -2 (= 254 as uint8)."""
    from pycusdr_amd.demodulator import STX
    bs = 14
    N = 1 << bs
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=4)
    conf['GPU']['UHF']['peakThresholdScale'] = 4.5
    p = loadProtocol('bench_GMSK')(conf=conf)
    gpu = STX.Demodulator(conf, p, 'UHF-H')
    with monkeypatch.context() as m:
        m.setattr(dbm, 'MFBank', OracleBank)
        cpu = STX.Demodulator(conf, loadProtocol('bench_GMSK')(conf=conf), 'UHF-H')
    x = sg.awgn(sg.get_padded_packet('GMSK')[0][30000:30000 + N], 15.0, rng=np.random.RandomState(3)).astype(np.complex64)
    x[5000] *= 60           # an interference spike inside the packet
    rg, rc = gpu.get_signalBufferHostPointer(), cpu.get_signalBufferHostPointer()
    rg[:] = x
    rc[:] = x
    og, oc = gpu.uploadAndFindCarrier(rg), cpu.uploadAndFindCarrier(rc)
    assert og[:2] == (0, 0) and list(og[2]) == list(oc[2]) == [5000]
    assert abs(rg[5000]) < 10 and np.array_equal(rg, rc)            # clipped in place, identically
    bg, cg, tg, sg_ = gpu.demodulate()
    bc, cc, tc, sc_ = cpu.demodulate()
    assert int(gpu.dopplerIdxlast) == N // 4 and sg_ == sc_
    assert np.count_nonzero(bg != bc) <= 1
    # trust bytes are raw bytes of fp32 magnitudes (quirk Q3), so a stray 254 can occur anywhere; the
    # tag itself is the run of >= 4 consecutive symbols around the clipped sample, same place in both
    def tagged_run(t):
        both = (t == 254).astype(int)
        runs = np.flatnonzero(np.convolve(both, np.ones(4, int), 'valid') == 4)
        return runs[0] if len(runs) else -1
    assert tagged_run(tg) >= 0 and tagged_run(tg) == tagged_run(tc)
    gpu.close()


def test_noise_reference_bin_through_the_demodulator(monkeypatch):
    """noise_measure_offset_Hz prepends a noise bin (reference DB:150-159): the pick skips it and the
    metric becomes the peak-to-noise-bin ratio."""
    bs = 14
    N = 1 << bs
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=16)
    conf['Radios']['Rx']['UHF-H']['noise_measure_offset_Hz'] = -38400
    p = loadProtocol('bench_GMSK')(conf=conf)
    gpu = UHF.Demodulator(conf, p, 'UHF-H')
    assert gpu.doppIdxArrayLen == 17 and gpu.doppIdxArrayOffset == 1
    with monkeypatch.context() as m:
        m.setattr(dbm, 'MFBank', OracleBank)
        cpu = UHF.Demodulator(conf, loadProtocol('bench_GMSK')(conf=conf), 'UHF-H')
    x = sg.awgn(sg.get_padded_packet('GMSK')[0][30000:30000 + N], 10.0, rng=np.random.RandomState(5)).astype(np.complex64)
    og, oc = gpu.uploadAndFindCarrier(x), cpu.uploadAndFindCarrier(x)
    assert int(gpu.dopplerIdxlast) == int(cpu.dopplerIdxlast)
    assert abs(og[1] - oc[1]) <= 1e-4 * abs(oc[1]) + 1e-6 and og[1] > 0          # peak / noise-bin metric
    assert abs(og[0] - oc[0]) < 0.5
    gpu.close()


def test_cc11xx_frame_received_and_crc_ok():
    """The same CC11xx end-to-end check as the CPU suite, on the HIP path, at the C5 geometry's
    samples per symbol (128) and a longer block."""
    from pycusdr_amd.demodulator_process import DemodulatorRunner
    from test_demod_hostlogic import _cc11xx_stimulus
    bs, sps = 17, 128
    conf = cfg.cc11xx_config(blockSize=bs, doppCarrierSteps=64, samplesPerSym=sps)
    p = loadProtocol('CC11xx')(conf=conf)
    p.CRC_CHECK = 'framer'        # the stimulus is a TX-framer frame: CRC inside the length-counted bytes
    run = DemodulatorRunner(conf, p, 'UHF-H')
    payload = np.arange(1, 41, dtype=np.uint8)
    sig = _cc11xx_stimulus(bs, sps, payload)
    step = (1 << bs) - (1 << 10)
    chunks = [sig[i * step:(i + 1) * step] for i in range(4)]      # 2136 bits must follow the header
    res, packets = run.run(chunks, decoder=Decoder({}, p))
    assert abs(res[0]['doppler']) < 600 and abs(res[0]['spSymEst'] - 128) < 1
    assert len(packets) == 1
    data, crc_err, _ = packets[0].getBinaryData()
    assert packets[0].packetLen == 42 and not crc_err and np.array_equal(data[:-2], payload)
    run.close()


def test_runner_stream_of_arbitrary_chunks_on_hip():
    """N1 on the HIP path: the BER bench's 2^14-sample chunks (and GNU Radio's 4095/4096) go through the ring
    buffer into blocks; results equal the block-fed run, the dict carries every reference key and pickles."""
    import pickle
    from pycusdr_amd.demodulator_process import DemodulatorRunner
    bs, ov = 15, 1 << 10
    N = 1 << bs
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=32)
    p = loadProtocol('bench_GMSK')(conf=conf)
    sig = sg.awgn(np.concatenate((sg.get_padded_packet('GMSK')[0], np.zeros(2 * N))), 12.0, rng=np.random.RandomState(4)).astype(np.complex64)
    step = N - ov
    nblk = len(sig) // step
    ref_run = DemodulatorRunner(conf, p, 'UHF-H')
    ref, _ = ref_run.run([sig[i * step:(i + 1) * step] for i in range(nblk)])
    ref_run.close()
    for chunk in (1 << 14, 4095):
        run = DemodulatorRunner(conf, p, 'UHF-H')
        got, packets = run.run_stream((sig[i:i + chunk] for i in range(0, nblk * step, chunk)), decoder=Decoder(conf, p))
        run.close()
        assert len(got) == nblk and len(packets) == 1 and packets[0].checkPacketData() == 0
        for a, b in zip(ref, got):
            assert np.array_equal(a['data'], b['data']) and np.array_equal(a['trust'], b['trust'])
            assert a['doppler'] == b['doppler'] and a['spSymEst'] == b['spSymEst']
            assert np.array_equal(np.float64(a['SNR']), np.float64(b['SNR']), equal_nan=True)
        d = pickle.loads(pickle.dumps(got[3], protocol=pickle.HIGHEST_PROTOCOL))
        assert {'rangerateEst', 'baudRate_est', 'baudrate_est', 'rangerate', 'data', 'trust', 'doppler', 'SNR'} <= set(d)
        assert np.array_equal(d['data'], got[3]['data'])


def test_ber_bench_script_runs_the_whole_chain():
    """examples/benchmark/bench_modem.py (the reference's BER bench, in-process): every packet found, error-free at high SNR."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples', 'benchmark', 'bench_modem.py')
    spec = importlib.util.spec_from_file_location('bench_modem', path)
    bm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bm)
    for mod, search in (('FSK', 'transforms'), ('GMSK', 'energy')):
        r = bm.run_snr(mod, 2, 14.0, 15, search, seed=5)
        assert r['packets'] == 2 and r['BER'] < 1e-3, r
    assert abs(bm.bandwidth('GMSK', 9600) - 9600 / 0.7) < 1e-9 and bm.bandwidth('FSK', 9600) == 28800


@pytest.mark.parametrize('first', [3, 4])
def test_one_runner_streams_twice(first):
    """A runner is a long-lived object: ``run_stream`` (and ``feed_device_begin``) called again go on in the page-locked
    buffer -- and behind the overlap -- the previous call ended in, whichever of the two that is (an odd number of blocks
    ends on the second one).  One stream cut in two calls, then continued block by block through the begin / end pair,
    equals the same stream in one call."""
    from pycusdr_amd.demodulator_process import DemodulatorRunner
    bs, ov = 15, 1 << 10
    N = 1 << bs
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=32)
    p = loadProtocol('bench_GMSK')(conf=conf)
    sig = sg.awgn(np.concatenate((sg.get_padded_packet('GMSK')[0], np.zeros(4 * N))), 12.0, rng=np.random.RandomState(4)).astype(np.complex64)
    step = N - ov
    nblk = len(sig) // step
    assert nblk >= first + 4
    one = DemodulatorRunner(conf, p, 'UHF-H')
    ref, ref_packets = one.run_stream([sig[:nblk * step]], decoder=Decoder(conf, p))
    one.close()
    run, dec = DemodulatorRunner(conf, p, 'UHF-H'), Decoder(conf, p)
    # (the one-block loop: its two page-locked buffers alternate block by block; `ref` above went through whatever the default
    # loop is -- batches of what the source has ready)
    got, packets = run.run_stream([sig[:first * step]], decoder=dec, blocks_per_call=1)
    assert run.raw is (run.demod.bank.input2 if first % 2 else run.demod.bank.input)
    more, pk = run.run_stream([sig[first * step:(first + 3) * step]], decoder=dec, blocks_per_call=1)
    got, packets = got + more, packets + pk
    for i in range(first + 3, nblk):          # ... and on through the two-halves call, one block in flight
        run.feed_device_begin(sig[i * step:(i + 1) * step])
        d = run.feed_host(run.feed_device_end())
        pk, _, d['numSyncSig'] = dec.findFrames(d['data'], 0)
        got.append(d)
        packets += pk
    run.close()
    assert [d['count'] for d in got] == list(range(nblk)) and len(packets) == len(ref_packets) == 1
    assert packets[0].checkPacketData() == 0
    for a, b in zip(ref, got):
        assert np.array_equal(a['data'], b['data']) and np.array_equal(a['trust'], b['trust'])
        assert a['doppler'] == b['doppler'] and a['spSymEst'] == b['spSymEst']
        assert np.array_equal(np.float64(a['SNR']), np.float64(b['SNR']), equal_nan=True)
