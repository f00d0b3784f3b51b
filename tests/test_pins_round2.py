"""Host functions pinned by fixtures recorded from the importable numpy parts of the reference
(tests/golden/make_golden.py, G9-G13): peak clipping, SNR estimate, FLAGS-mode packet search, CC11xx
packet parsing, and the soft combiner's alignment cross-correlation (oracle restatement)."""
import types

import numpy as np
import pytest

from oracle import mfbank_oracle as orc
from pycusdr_amd import config as cfg
from pycusdr_amd.decoder import Decoder
from pycusdr_amd.demodulator.demodulator_base import Demodulator
from pycusdr_amd.protocol import PacketEndDetect, loadProtocol

G9_CASES = [(n, s) for n in ('none', 'one', 'close', 'far', 'loud') for s in (4.5, 40.5)]


@pytest.mark.parametrize('name,scale', G9_CASES)
def test_threshold_input_matches_reference(goldens, name, scale):
    """_thresholdInput == the reference's Demodulator.__thresholdInput (DB:670-707): clipped samples,
    clippedPeakIPure, clippedPeakI (gaps below 100 samples filled, larger gaps kept)."""
    k = f'g9/{name}/s{scale}'
    s = types.SimpleNamespace(peakThresholdScale=scale, Nfft=len(goldens[k + '/in']))
    y = goldens[k + '/in'].copy()
    Demodulator._thresholdInput(s, y)
    assert np.array_equal(y, goldens[k + '/out'])
    assert np.array_equal(np.asarray(s.clippedPeakIPure), goldens[k + '/clippedPeakIPure'])
    assert np.array_equal(np.asarray(s.clippedPeakI), goldens[k + '/clippedPeakI'])


def test_compute_snr_matches_reference(goldens):
    """computeSNR == the reference's (DB:635-667), signal or noise band wrapping around bin 0 included; the
    spectrum windows come through the same get_spectrum(start, count) call the HIP path serves."""
    X, shifts = goldens['g10/X'], goldens['g10/shifts']
    N = len(X)
    bank = types.SimpleNamespace(get_spectrum=lambda start, count: X[(start + np.arange(count)) % N])
    s = types.SimpleNamespace(doppCyperSymNorm=shifts, Nfft=N, bank=bank)
    s._spectrum_slice = lambda a, b: Demodulator._spectrum_slice(s, a, b)
    got = np.array([Demodulator.computeSNR(s, int(lo), int(hi), int(w)) for lo, hi, w in goldens['g10/pairs']])
    assert np.array_equal(np.isnan(got), np.isnan(goldens['g10/snr']))
    ok = ~np.isnan(got)
    assert np.allclose(got[ok], goldens['g10/snr'][ok], rtol=1e-6, atol=1e-6)
    # and the oracle's statement of the same function
    ref = np.array([orc.compute_snr(X, shifts, int(lo), int(hi), int(w), N) for lo, hi, w in goldens['g10/pairs']])
    assert np.allclose(ref[ok], goldens['g10/snr'][ok], rtol=1e-6, atol=1e-6)


def run_flags_kat(goldens, name, correlator):
    p = loadProtocol('bench_GMSK')(conf=cfg.bench_config())
    p.packetEndDetectMode = PacketEndDetect.FLAGS
    d = Decoder({}, p, correlator=correlator) if correlator is not None else Decoder({}, p)     # None: the HIP correlator
    stream = goldens[f'g11/{name}/stream'].astype(np.float64)
    cuts = goldens[f'g11/{name}/cuts']
    seen = 0
    for ci in range(len(cuts) - 1):
        packets, bits, nsync = d.findFrames(stream[cuts[ci]:cuts[ci + 1]], 1000 * ci)
        k = f'g11/{name}/call{ci}'
        assert nsync == int(goldens[k + '/numSyncSig'])
        assert len(packets) == int(goldens[k + '/npackets'])
        assert (-1 if d.headerFrameStartIdx is None else d.headerFrameStartIdx) == int(goldens[k + '/pending'])
        for i, q in enumerate(packets):
            assert np.array_equal(np.asarray(q.bits).astype(np.uint8), goldens[k + f'/p{i}/bits'])
            assert q.frameStartIdx == int(goldens[k + f'/p{i}/start'])
            assert q.maskBitErrors == float(goldens[k + f'/p{i}/maskBitErrors'])
            assert q.frameSplitIdx == int(goldens[k + f'/p{i}/split'])
            seen += 1
    return seen


@pytest.mark.parametrize('name', ['inside', 'across', 'overflow', 'two'])
def test_flags_mode_findframes_matches_reference(goldens, name):
    """FLAGS-mode packet search (DEC:122-243): frame inside one call, across calls (frameSplitIdx), forced end
    past maxPacketLenBits, two frames in one call."""
    assert run_flags_kat(goldens, name, orc.sync_correlate) >= 1


@pytest.mark.parametrize('name', ['framer', 'hardware', 'rawcrc', 'rawcrc_bad'])
def test_cc11xx_packet_matches_reference(goldens, name):
    """PacketCC11xx: length byte, cut, PN9 de-whitening and -- by default -- the reference's CRC flag, bit for
    bit (CC11xx.py:226-300).  The flag of the reference reads 'no error' only for a frame whose two bytes behind
    the data are the RAW crc16([len | data]); frames of its own TX framer and hardware-mode frames both read
    'error'.  CRC_CHECK = 'framer' is the opt-in check at the framer's CRC position."""
    p = loadProtocol('CC11xx')(conf=cfg.cc11xx_config())
    k = f'g12/{name}'
    pk = p.Packet(goldens[k + '/bits_in'].astype(np.float64), 5, 1.0)
    assert int(pk.packetLen) == int(goldens[k + '/packetLen'])
    assert np.array_equal(np.asarray(pk.bits).astype(np.uint8), goldens[k + '/bits_cut'])
    data, flag, again = pk.getBinaryData()
    assert np.array_equal(data, goldens[k + '/data']) and bool(flag) == bool(goldens[k + '/flag'])
    p.CRC_CHECK = 'framer'
    data2, err, _ = p.Packet(goldens[k + '/bits_in'].astype(np.float64), 5, 1.0).getBinaryData()
    assert np.array_equal(data2, data)
    assert bool(err) == (name != 'framer')      # only the TX framer's layout carries its CRC inside the counted bytes


def test_cc11xx_gfsk2_flavour_selectable(goldens):
    """The reference's modIDX = 1 build (CC11xx over GFSK2, CC11xx.py:30-32,42) is a registry entry here; its
    filter bank is GFSK2.get_filter (pinned by G1 'GFSK2_sps16'), the framing is shared."""
    p = loadProtocol('CC11xx_GFSK2')(conf=cfg.cc11xx_config())
    M, bank = p.get_filter(1 << 10, 16, 3)
    assert M == 8 and np.array_equal(bank, goldens['g1/GFSK2_sps16/bank_n1024'])
    q = loadProtocol('CC11xx')(conf=cfg.cc11xx_config())
    assert np.array_equal(p.get_mask(), q.get_mask()) and p.packetLen == q.packetLen and p.name == 'CC11xx GFSK-2'


@pytest.mark.parametrize('tag', ['r1_vs_d1', 'r2_vs_d2_16k', 'r1_shifted'])
def test_oracle_custom_xcorr_matches_reference(goldens, tag):
    """oracle.custom_xcorr == lib.customXCorr.customXCorr on the reference's own soft-combiner test streams."""
    a, b = goldens[f'g13/{tag}/a'].astype(np.float64), goldens[f'g13/{tag}/b'].astype(np.float64)
    r = orc.custom_xcorr(a, b)
    assert np.allclose(r[:64], goldens[f'g13/{tag}/head'], rtol=1e-12, atol=1e-9)
    assert np.allclose(np.abs(r), goldens[f'g13/{tag}/abs_c64'], rtol=1e-5, atol=1e-3)
    assert np.array_equal(np.argsort(-np.abs(r), kind='stable')[:3], goldens[f'g13/{tag}/top15_idx'][:3])


@pytest.mark.parametrize('name', ['grc', 'bench', 'odd', 'tight'])
def test_ring_buffer_matches_reference(goldens, name):
    """RingBuffer == the reference's sigFIFO.RingBuffer (sigFIFO.py:13-103) on seeded insert/pop traces: GNU
    Radio sized chunks (4095/4096), the BER bench's 2^14, odd sizes with wrap-around, the overflow flush."""
    from pycusdr_amd.sigFIFO import RingBuffer
    k = f'g14/{name}'
    rb = RingBuffer(int(goldens[k + '/outLen']), bufLen=int(goldens[k + '/bufLen']), dtype=np.complex64)
    v, pos = 0, 0
    for i, n in enumerate(goldens[k + '/chunks']):
        n = int(n)
        data = (np.arange(v, v + n) + 1j * (np.arange(v, v + n) % 7)).astype(np.complex64)
        v += n
        assert rb.insert(data) == int(goldens[k + '/sizes'][i])
        blk = rb.popBlock(rb.outLen)
        assert len(blk) == int(goldens[k + '/npop'][i])
        assert np.array_equal(np.asarray(blk, dtype=np.complex64), goldens[k + '/popped'][pos:pos + len(blk)])
        pos += len(blk)
        assert (rb.headIdx, rb.tailIdx, rb.currentBufSize) == tuple(int(q) for q in goldens[k + '/state'][i])
    with pytest.raises(IndexError):
        RingBuffer(100, bufLen=50)
    # overflow: the reference raises ValueError after its flush (stale end index, fixture records it); here the
    # flush is followed by a clean store
    assert bool(goldens['g14/overflow_raises_in_reference'])
    rb = RingBuffer(1900, bufLen=2000, dtype=np.complex64)
    rb.insert(np.zeros(900, np.complex64))
    rb.insert(np.ones(900, np.complex64))
    assert rb.insert(np.full(900, 2, np.complex64)) == 900 and np.all(rb.buf[:900] == 2)


def test_sigfifo_blocks_from_arbitrary_chunks():
    """SigFIFO.getBlock: the stream comes out unchanged, cut into fixed blocks, whatever the chunk size."""
    from pycusdr_amd.sigFIFO import SigFIFO
    rs = np.random.RandomState(3)
    x = (rs.standard_normal(70000) + 1j * rs.standard_normal(70000)).astype(np.complex64)
    for chunk in (16384, 4095, 1000, 15360, 20000):
        fifo = SigFIFO((x[i:i + chunk] for i in range(0, len(x), chunk)), 15360)
        got = []
        with pytest.raises(TimeoutError):
            while True:
                got.append(np.array(fifo.getBlock()))
        assert len(got) == len(x) // 15360 and np.array_equal(np.concatenate(got), x[:len(got) * 15360])


def test_oversized_chunks_and_block_assembler():
    """Chunks larger than the ring (2 blocks) go through SigFIFO piece by piece; the ring itself refuses what it cannot hold
    (the reference raises ValueError there too); BlockAssembler builds the same blocks in place with the overlap carried."""
    from pycusdr_amd.sigFIFO import BlockAssembler, RingBuffer, SigFIFO
    rs = np.random.RandomState(8)
    x = (rs.standard_normal(200000) + 1j * rs.standard_normal(200000)).astype(np.complex64)
    for chunk in (65536, 31745, 30720, 100000):
        fifo = SigFIFO((x[i:i + chunk] for i in range(0, len(x), chunk)), 15360)
        got = []
        with pytest.raises(TimeoutError):
            while True:
                got.append(np.array(fifo.getBlock()))
        assert len(got) == len(x) // 15360 and np.array_equal(np.concatenate(got), x[:len(got) * 15360])
    with pytest.raises(ValueError):
        RingBuffer(100, bufLen=200).insert(np.zeros(201, np.complex64))
    N, ov = 4096, 256
    for chunk in (1000, 4096, 3840, 9001, 17):
        buf = np.zeros(N, np.complex64)
        buf[:ov] = x[:ov]
        asm = BlockAssembler(buf, ov)
        nb = 0
        for i in range(ov, 60000, chunk):
            for blk in asm.push(x[i:min(i + chunk, 60000)]):
                assert blk is buf and np.array_equal(blk, x[nb * (N - ov): nb * (N - ov) + N])
                nb += 1
        assert nb == asm.blocks == (60000 - ov) // (N - ov)
    with pytest.raises(IndexError):
        BlockAssembler(np.zeros(8, np.complex64), 8)
