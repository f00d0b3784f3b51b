"""mfb_receive_blocks_begin / _end -- B consecutive blocks of the stream per device call -- against the one-block call
(mfb_receive_block): every number of every block must be identical, bit for bit, whatever B is, for all four modulations,
with an all-zero (NaN pick) block inside a batch and a symbol slip at a batch edge.  Reference: the loop these calls batch is
Demodulator_process.run, demodulator_process.py:284-338 (one block per turn); the per-block arithmetic is DB:548-632, 711-1009."""
import copy

import numpy as np
import pytest

from pycusdr_amd import config as cfg, signals as sg
from pycusdr_amd.decoder import Decoder
from pycusdr_amd.demodulator import UHF
from pycusdr_amd.demodulator_process import DemodulatorRunner
from pycusdr_amd.protocol import loadProtocol

pytestmark = pytest.mark.gpu


def _one(conf):
    """The configuration with ``"blocks_per_call": 1``: the reference's loop, one block per turn (DP:284-338) -- what the batched
    loops are held against.  (Without the key ``run_stream`` batches whatever its source has ready.)"""
    c = copy.deepcopy(conf)
    c['GPU']['UHF'].setdefault('HIP', {})['blocks_per_call'] = 1
    return c


def _same(a, b):
    return bool(np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True))


def _blocks_one_by_one(demod, sig, nblocks, step, N):
    out = []
    for b in range(nblocks):
        raw = demod.get_signalBufferHostPointer()
        raw[:] = sig[b * step: b * step + N]
        est = demod.uploadAndFindCarrier(raw)
        scores = demod.bank.get_scores().copy()
        out.append((est, demod.demodulateDevice(), scores))
    return out


def _check_block(tag, got, want):
    (ea, ra), (eb, rb, _) = got, want
    assert _same([ea[0], ea[1], ea[3]], [eb[0], eb[1], eb[3]]), (tag, ea, eb)
    assert ra['spSym'] == rb['spSym'], tag
    for k in ('symbols', 'centres', 'trust'):
        assert _same(ra[k], rb[k]), (tag, k)


@pytest.mark.parametrize('mod,pname,bs,D', [('GMSK', 'bench_GMSK', 15, 64), ('FSK', 'bench_FSK', 16, 33), ('GFSK', 'bench_GFSK', 14, 16),
                                            ('BPSK', 'bench_BPSK', 15, 40)])
def test_batches_equal_the_one_block_call(mod, pname, bs, D):
    N, ov = 1 << bs, 1 << 10
    step = N - ov
    nblocks = 11
    conf = cfg.bench_config(pname, blockSize=bs, doppCarrierSteps=D)
    p = loadProtocol(pname)(conf=conf)
    sig = sg.s1_stream(nblocks, N, ov, mod, snr_db=9.0, seed=11)
    sig[3 * step: 4 * step + ov] = 0        # an all-zero block (NaN index, skipped: DB:625-630) inside every batch size below
    one, bat = UHF.Demodulator(conf, p, 'UHF-H'), UHF.Demodulator(conf, p, 'UHF-H')
    try:
        want = _blocks_one_by_one(one, sig, nblocks, step, N)
        assert want[3][0][0] == 0.0 and want[3][0][3] == 0.0       # the zero block was skipped
        for B in (2, 5, 8):
            wins = bat.blockWindows(B)
            b0, turn = 0, 0
            while b0 < nblocks:
                nb = min(B, nblocks - b0)
                w = wins[turn & 1]
                w[:nb * step + ov] = sig[b0 * step: (b0 + nb) * step + ov]
                bat.beginBlocks(turn & 1, nb, source=('window', 'window2')[turn & 1])
                got = bat.endBlocks(turn & 1)
                assert len(got) == nb
                for i in range(nb):
                    _check_block((B, b0 + i), got[i], want[b0 + i])
                b0 += nb
                turn += 1
    finally:
        one.close()
        bat.close()


def test_batch_scores_equal_one_block_scores_bit_for_bit():
    """The score table of every block of a batch against the one-block search (the numbers behind the pick): read from the
    batch's device table through a second batch of ONE block at each position."""
    bs, D, ov = 15, 48, 1 << 10
    N = 1 << bs
    step = N - ov
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=D)
    p = loadProtocol('bench_GMSK')(conf=conf)
    sig = sg.s1_stream(6, N, ov, 'GMSK', snr_db=6.0, seed=3)
    one, bat = UHF.Demodulator(conf, p, 'UHF-H'), UHF.Demodulator(conf, p, 'UHF-H')
    try:
        want = _blocks_one_by_one(one, sig, 6, step, N)
        w = bat.blockWindows(6)[0]
        w[:] = sig[:6 * step + ov]
        bat.beginBlocks(0, 6)
        got = bat.endBlocks(0)
        for i in range(6):
            _check_block(i, got[i], want[i])
            # the pick's two floats are functions of the whole table: equal picks on different tables would be a coincidence;
            # the metric is the weighted score itself
            assert got[i][0][1] == want[i][0][1]
    finally:
        one.close()
        bat.close()


def test_long_filters_and_noise_bin_in_a_batch():
    """CC11xx geometry (384-tap filters: 2048-point segments, one wave per segment; IF offset; noise-reference bin)."""
    from pycusdr_amd.protocol.CC11xx import frame_bits
    bs, sps = 17, 128
    N = 1 << bs
    step = N - 1024
    conf = cfg.cc11xx_config(blockSize=bs, doppCarrierSteps=48, samplesPerSym=sps)
    conf['Radios']['Rx']['UHF-H']['noise_measure_offset_Hz'] = 300000
    p = loadProtocol('CC11xx')(conf=conf)
    rs = np.random.RandomState(4)
    fs = 7416 * sps
    bits = np.concatenate([frame_bits(rs.randint(0, 256, 60).astype(np.uint8), preamble=(0xAA,) * 10) for _ in range(6)])
    sig = sg.modulateFSK(bits, sps)
    sig = np.tile(sig, int(np.ceil((3 * step + 1024) / len(sig))))[:3 * step + 1024]
    sig = sg.awgn(sig * np.exp(2j * np.pi * 148320 / fs * np.arange(len(sig))), 12.0, rng=np.random.RandomState(2)).astype(np.complex64)
    one, bat = UHF.Demodulator(conf, p, 'UHF-H'), UHF.Demodulator(conf, p, 'UHF-H')
    try:
        want = _blocks_one_by_one(one, sig, 3, step, N)
        w = bat.blockWindows(3)[1]
        w[:] = sig
        bat.beginBlocks(1, 3, source='window2')
        got = bat.endBlocks(1)
        for i in range(3):
            _check_block(i, got[i], want[i])
            assert len(got[i][1]['symbols']) > 900
    finally:
        one.close()
        bat.close()


@pytest.mark.parametrize('B', [2, 5, 8])
def test_batched_stream_equals_one_block_stream(B):
    """run_stream with "HIP": {"blocks_per_call": B}: result dicts, bits, trust and packets equal the one-block loop's, with a
    symbol slip planted in the first block of a batch (the alignment against the last block of the previous batch, DB:863-988)
    and chunks that do not divide the window."""
    bs, ov = 15, 1 << 10
    N = 1 << bs
    step = N - ov
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=32)
    p = loadProtocol('bench_GMSK')(conf=conf)
    nblocks = 2 * B + 3
    sig = sg.s1_stream(nblocks, N, ov, 'GMSK', snr_db=10.0, seed=5)[ov:]
    confB = copy.deepcopy(conf)
    confB['GPU']['UHF'].setdefault('HIP', {})['blocks_per_call'] = B
    # the chunk -> window copies: on the spot for writable chunks (B = 2), on the library's copy thread for read-only ones
    # (B = 5: a recording) or for all of them when the configuration says so (B = 8)
    if B == 5:
        sig.flags.writeable = False
    elif B == 8:
        confB['GPU']['UHF']['HIP']['async_copies'] = True
    a, b = DemodulatorRunner(_one(conf), p, 'UHF-H'), DemodulatorRunner(confB, p, 'UHF-H')

    def slipping(run):
        """Drop one symbol decision at the front of block B (first block of the second batch) -- what a +1 slip of the
        symbol clock at a block edge looks like to checkSymbolOverlap."""
        inner = run.demod.demodulateHost

        def host(rec, prev_tail=None):
            if host.calls == B:
                rec = dict(rec)
                for k in ('symbols', 'centres', 'trust'):
                    rec[k] = rec[k][1:]
                rec.pop('_bits', None)
                rec.pop('_a13', None)          # (the device's own alignment of the unmodified block)
            host.calls += 1
            return inner(rec, prev_tail=prev_tail)
        host.calls = 0
        run.demod.demodulateHost = host
    try:
        slipping(a)
        slipping(b)
        da, db = Decoder(conf, p), Decoder(conf, p)
        ra, pa = a.run_stream((sig[i:i + 16384] for i in range(0, len(sig), 16384)), decoder=da)
        rb, pb = b.run_stream((sig[i:i + 5000] for i in range(0, len(sig), 5000)), decoder=db)
        assert len(ra) == len(rb) == nblocks
        for x, y in zip(ra, rb):
            assert x['count'] == y['count']
            for k in ('doppler', 'doppler_std', 'SNR', 'spSymEst', 'baudrate_est', 'rangerate', 'numSyncSig'):
                assert _same(x[k], y[k]), (x['count'], k)
            assert _same(x['data'], y['data']) and _same(x['trust'], y['trust']), x['count']
            assert 'latency_ms' in y and y['rate_ksps'] > 0
        assert len(pa) == len(pb) and all(_same(u.bits, v.bits) for u, v in zip(pa, pb))
        assert _same(a.demod.poswinP, b.demod.poswinP) and _same(a.demod.posSymEnd, b.demod.posSymEnd)
        assert (getattr(b, '_copier', None) is not None) and not b._copier._held            # the copy thread exists and holds nothing back
        # the device did the bit lookup and the alignment of (nearly) every block, and the decoder's searches came with them: all
        # but the block with the planted slip -- which goes through the host code -- and what was in flight behind it
        assert b.demod.stage_blocks >= nblocks - 2 * B - 1 and db.ahead_blocks >= nblocks - 2 * B - 1, (b.demod.stage_blocks, db.ahead_blocks)
        # a second stream on the same runners goes on behind the overlap the first left
        more = sg.s1_stream(B + 1, N, ov, 'GMSK', snr_db=10.0, seed=6)[ov:]
        ra2, _ = a.run_stream([more], decoder=da)
        rb2, _ = b.run_stream([more], decoder=db)
        assert len(ra2) == len(rb2) == B + 1
        assert all(_same(x['data'], y['data']) for x, y in zip(ra2, rb2))
    finally:
        a.close()
        b.close()


def test_more_blocks_per_call_than_the_device_stages_take():
    """70 blocks per call (the device's stream stages chain at most 64 blocks, a recorded graph is kept per size up to 32): the
    batch runs, the integer stages fall to the host code, everything equals the one-block loop."""
    bs, ov, B = 13, 1 << 10, 70
    N = 1 << bs
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=16)
    p = loadProtocol('bench_GMSK')(conf=conf)
    nblocks = 2 * B + 9
    sig = sg.s1_stream(nblocks, N, ov, 'GMSK', snr_db=12.0, seed=9)[ov:]
    confB = copy.deepcopy(conf)
    confB['GPU']['UHF'].setdefault('HIP', {})['blocks_per_call'] = B
    a, b = DemodulatorRunner(_one(conf), p, 'UHF-H'), DemodulatorRunner(confB, p, 'UHF-H')
    try:
        da, db = Decoder(conf, p), Decoder(conf, p)
        ra, pa = a.run_stream([sig], decoder=da)
        rb, pb = b.run_stream((sig[i:i + 30000] for i in range(0, len(sig), 30000)), decoder=db)
        assert len(ra) == len(rb) == nblocks
        for x, y in zip(ra, rb):
            for k in ('count', 'doppler', 'doppler_std', 'SNR', 'spSymEst', 'numSyncSig'):
                assert _same(x[k], y[k]), (x['count'], k)
            assert _same(x['data'], y['data']) and _same(x['trust'], y['trust']), x['count']
        assert len(pa) == len(pb) and all(_same(u.bits, v.bits) for u, v in zip(pa, pb))
        assert getattr(b.demod, 'stage_blocks', 0) <= 9        # only the short last batch could run its stages on the device
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize('with_decoder', [True, False])
def test_a_live_source_gets_every_complete_block_out_at_its_markers(with_decoder):
    """A source that yields ``None`` where it would block ("nothing more right now", ``DemodulatorRunner.drain_marked``): the
    complete blocks go out as a shorter batch at every marker -- batches of 0 ... B blocks in one stream, one recorded graph per
    batch size -- and nothing waits for a full window.  Same dicts, bits and packets as the one-block loop; the one-block loop
    and the pipelined form skip the markers."""
    bs, ov, B = 15, 1 << 10, 6
    N = 1 << bs
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=32)
    p = loadProtocol('bench_GMSK')(conf=conf)
    nblocks = 31
    sig = sg.s1_stream(nblocks, N, ov, 'GMSK', snr_db=10.0, seed=8)[ov:]
    rs = np.random.RandomState(4)
    cuts = np.sort(rs.choice(np.arange(1, len(sig) // 4096), 14, replace=False)) * 4096
    chunks = [sig[i:i + 4096] for i in range(0, len(sig), 4096)]

    def live():
        """The stream in 4096-sample chunks through drain_marked: the transport runs dry at 14 random places."""
        DRY = object()
        items = []
        for i, c in enumerate(chunks):
            if i * 4096 in cuts:
                items.append(DRY)
            items.append(c)
        items.reverse()

        def wait():
            while items and items[-1] is DRY:
                items.pop()
            return items.pop() if items else None

        def poll():
            if items and items[-1] is DRY:
                items.pop()
                return None
            return wait()
        return DemodulatorRunner.drain_marked(poll, wait)

    confB = copy.deepcopy(conf)
    confB['GPU']['UHF'].setdefault('HIP', {})['blocks_per_call'] = B
    a, b = DemodulatorRunner(conf, p, 'UHF-H'), DemodulatorRunner(confB, p, 'UHF-H')
    try:
        da, db = (Decoder(conf, p), Decoder(conf, p)) if with_decoder else (None, None)
        ra, pa = a.run_stream(live(), decoder=da, blocks_per_call=1)    # one-block loop: markers collect the block in flight
        sizes = []
        inner = b.demod.beginBlocks
        b.demod.beginBlocks = lambda which, nb, **kw: (sizes.append(nb), inner(which, nb, **kw))[1]
        delivered_at_marker = []
        out = []

        def marked():
            for c in live():
                if c is None:
                    yield None
                    delivered_at_marker.append(len(out))               # what had been delivered when the loop came back for more
                else:
                    yield c
        rb, pb = b.run_stream(marked(), decoder=db, sink=out.append)
        rb = out
        assert len(ra) == len(rb) == nblocks
        for x, y in zip(ra, rb):
            assert x['count'] == y['count']
            for k in ('doppler', 'doppler_std', 'SNR', 'spSymEst', 'baudrate_est', 'rangerate') + (('numSyncSig',) if with_decoder else ()):
                assert _same(x[k], y[k]), (x['count'], k)
            assert _same(x['data'], y['data']) and _same(x['trust'], y['trust']), x['count']
        assert len(pa) == len(pb) and all(_same(u.bits, v.bits) for u, v in zip(pa, pb))
        assert sum(sizes) == nblocks and max(sizes) <= B and len(set(sizes)) >= 3, sizes
        # at every marker every block that was complete had been delivered: nothing is held back for a fuller window
        step = N - ov
        for cut, n in zip(cuts, delivered_at_marker):
            assert n == cut // step, (cut, n)
        # the pipelined form takes the same source
        rc, _ = a.run_stream(live(), pipelined=True)
        assert len(rc) == nblocks
        # a marked source gets batches without being configured (they cost it no latency): 2^15-sample blocks -> up to 32 per call
        assert a.auto_blocks_per_call() == 32 and a.blocks_per_call() is None
        seen = []
        inner_a = a.demod.beginBlocks
        a.demod.beginBlocks = lambda which, nb, **kw: (seen.append(nb), inner_a(which, nb, **kw))[1]
        rd, _ = a.run_stream(live())
        assert len(rd) == nblocks and sum(seen) == nblocks and max(seen) > 1
        # ... and so does a plain iterator whose chunks are simply there (a recording): windows of up to 32 blocks, same results
        del seen[:]
        re_, _ = a.run_stream(iter(chunks))
        assert len(re_) == nblocks and sum(seen) == nblocks and max(seen) > 1, seen
        # "blocks_per_call": 1 (or the argument) keeps the reference's one-block loop
        del seen[:]
        rf, _ = a.run_stream(iter(chunks), blocks_per_call=1)
        assert len(rf) == nblocks and not seen
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize('overlap', [False, True])
def test_a_batch_beside_the_oracle_directly(monkeypatch, overlap):
    """Five blocks of 2^15 samples x 64 bins through ONE batched call (mfb_receive_blocks_*), every block held to the ORACLE -- not
    to the one-block call: the whole doppSum table of every block against ``orc.doppler_scores`` (north_star's 1e-5), the pick
    against ``orc.find_doppler_est``, and the bits, centres and alignment state against the same host driver over the CPU
    oracle bank, block after block.  Both arrangements of a batch on the device (one stream, two streams)."""
    import pycusdr_amd.demodulator.demodulator_base as dbm
    from oracle import mfbank_oracle as orc
    from oracle_bank import OracleBank
    from pycusdr_amd.demodulator import UHF
    bs, ov, D, B = 15, 1 << 10, 64, 5
    N = 1 << bs
    step = N - ov
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=D)
    p = loadProtocol('bench_GMSK')(conf=conf)
    gpu = UHF.Demodulator(conf, p, 'UHF-H')
    with monkeypatch.context() as m:
        m.setattr(dbm, 'MFBank', OracleBank)
        cpu = UHF.Demodulator(conf, loadProtocol('bench_GMSK')(conf=conf), 'UHF-H')
    assert type(gpu.bank).__name__ == 'MFBank' and isinstance(cpu.bank, OracleBank)
    sig = sg.s1_stream(B, N, ov, 'GMSK', snr_db=10.0, seed=31)
    try:
        gpu.bank.set_batch_overlap(overlap)
        win = gpu.blockWindows(B)[0]
        win[:] = sig[:B * step + ov]
        gpu.beginBlocks(0, B)
        recs = gpu.endBlocks(0)
        assert len(recs) == B
        for b, (est, rec) in enumerate(recs):
            x = sig[b * step: b * step + N]
            oc = cpu.uploadAndFindCarrier(x.copy())
            ref = cpu.bank.get_scores().astype(np.float64)
            got = gpu.bank.get_batch_scores(b).astype(np.float64)
            assert got.shape == ref.shape and np.abs(got - ref).max() / ref.max() < 1e-5, b       # correlation magnitudes
            # the pick: the oracle's scan over the DEVICE's table gives the frequency the batch reported, bit for bit
            oidx, _ = orc.find_doppler_est(gpu.bank.get_batch_scores(b), gpu.num_dopplers, gpu.doppIdxArrayOffset, True)
            mine = orc.interpolate_doppler(oidx, gpu.doppCyperSymNorm, gpu.doppHzLUT, gpu.centreFreqOffset)
            assert mine['freqOffset'] == est[0], (b, mine['freqOffset'], est[0])
            assert abs(est[0] - oc[0]) <= 1e-3 * max(1.0, abs(oc[0])) + 0.05 and abs(est[3] - oc[3]) < 1e-3     # Hz, SNR dB
            bg, cg, tg, spg = gpu.demodulateHost(rec)
            bc, cc, tc, spc = cpu.demodulate()
            assert spg == spc and len(bg) == len(bc) > 0.9 * N / 16, b
            assert np.array_equal(bg, bc), f'block {b}: {np.count_nonzero(np.asarray(bg) != np.asarray(bc))} symbol decisions differ'
            # a matched-filter peak that ties between neighbouring samples to fp32-vs-fp64 round-off may sit one sample off
            dc = (np.asarray(cg).astype(np.int16) - np.asarray(cc).astype(np.int16) + 128) % 256 - 128
            assert np.abs(dc).max() <= 1 and np.count_nonzero(dc) <= 2, (b, np.count_nonzero(dc))
            assert np.array_equal(np.asarray(gpu.poswinP), np.asarray(cpu.poswinP)), b
    finally:
        gpu.close()


@pytest.mark.parametrize('pname,mod,bs,D,B', [('bench_GMSK', 'GMSK', 15, 32, 5), ('bench_BPSK', 'BPSK', 14, 16, 3)])
def test_a_batch_on_two_streams_changes_no_number(pname, mod, bs, D, B):
    """``"HIP": {"batch_overlap": true}`` (mfb_set_batch_overlap): part 2 of a batch -- matched filters at the picked shift, rate,
    centres, the integer stages, the read-back -- on a second stream beside the next batch's search.  Every result dict, packet and
    the alignment state equal the one-stream loop's, and the handle goes back and forth between the two arrangements."""
    N, ov = 1 << bs, 1 << 10
    conf = cfg.bench_config(pname, blockSize=bs, doppCarrierSteps=D)
    p = loadProtocol(pname)(conf=conf)
    nblocks = 4 * B + 2
    sig = sg.s1_stream(nblocks, N, ov, mod, snr_db=10.0, seed=21)[ov:]
    confA, confB = copy.deepcopy(conf), copy.deepcopy(conf)
    confA['GPU']['UHF'].setdefault('HIP', {})['blocks_per_call'] = B
    confB['GPU']['UHF'].setdefault('HIP', {}).update(blocks_per_call=B, batch_overlap=True)
    a, b = DemodulatorRunner(confA, p, 'UHF-H'), DemodulatorRunner(confB, p, 'UHF-H')
    try:
        da, db = Decoder(conf, p), Decoder(conf, p)
        ra, pa = a.run_stream((sig[i:i + 9000] for i in range(0, len(sig), 9000)), decoder=da)
        rb, pb = b.run_stream((sig[i:i + 9000] for i in range(0, len(sig), 9000)), decoder=db)
        assert len(ra) == len(rb) == nblocks
        for x, y in zip(ra, rb):
            for k in ('doppler', 'doppler_std', 'SNR', 'spSymEst', 'numSyncSig'):
                assert _same(x[k], y[k]), (x['count'], k)
            assert _same(x['data'], y['data']) and _same(x['trust'], y['trust']), x['count']
        assert len(pa) == len(pb) and all(_same(u.bits, v.bits) for u, v in zip(pa, pb))
        assert _same(a.demod.poswinP, b.demod.poswinP) and _same(a.demod.posSymEnd, b.demod.posSymEnd)
        # the device did the integer stages on the second stream too (all but the blocks the noiseless padding makes irregular,
        # and what was in flight behind them), exactly as often as on one stream
        assert b.demod.stage_blocks == a.demod.stage_blocks >= nblocks // 2, (a.demod.stage_blocks, b.demod.stage_blocks)
        # the same handle, back to one stream and on to two again: a second stream of samples, still equal
        more = sg.s1_stream(2 * B + 1, N, ov, mod, snr_db=10.0, seed=22)[ov:]
        b.conf['GPU']['UHF']['HIP']['batch_overlap'] = False
        ra2, _ = a.run_stream([more], decoder=da)
        rb2, _ = b.run_stream([more], decoder=db)
        b.conf['GPU']['UHF']['HIP']['batch_overlap'] = True
        ra3, _ = a.run_stream([more], decoder=da)
        rb3, _ = b.run_stream([more], decoder=db)
        for u, v in zip(ra2 + ra3, rb2 + rb3):
            assert _same(u['data'], v['data']) and _same(u['SNR'], v['SNR']) and _same(u['trust'], v['trust'])
        # a batch in flight refuses the switch
        wins = b.demod.blockWindows(B)
        wins[0][:] = 0
        b.demod.beginBlocks(0, 2)
        with pytest.raises(Exception):
            b.demod.bank.set_batch_overlap(False)
        b.demod.endBlocks(0)
    finally:
        a.close()
        b.close()


def test_a_plain_iterator_gets_batches_while_it_has_a_backlog_and_single_blocks_while_it_is_live():
    """Nothing configured, no markers: ``run_stream`` decides from how long each chunk took to come.  A source that hands over a
    backlog (chunks that are there at once, one of them several blocks long) gets them in batches; the same source turned live
    (its chunks come in bursts of two that have to be waited for) gets every block out as soon as its last chunk is in -- delivered
    before the next burst is asked for -- and everything equals the one-block loop."""
    import time
    bs, ov = 13, 1 << 10
    N = 1 << bs
    step = N - ov
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=16)
    p = loadProtocol('bench_GMSK')(conf=conf)
    nblocks = 26
    sig = sg.s1_stream(nblocks, N, ov, 'GMSK', snr_db=12.0, seed=11)[ov:]
    backlog = 14 * step + 1000                        # the first 14 blocks and a bit are there at once
    a, b = DemodulatorRunner(_one(conf), p, 'UHF-H'), DemodulatorRunner(conf, p, 'UHF-H')
    try:
        ra, _ = a.run_stream([sig])
        sizes, out, asked = [], [], []
        inner = b.demod.beginBlocks
        b.demod.beginBlocks = lambda which, nb, **kw: (sizes.append(nb), inner(which, nb, **kw))[1]

        def source():
            yield sig[:5 * step]                      # one chunk of five blocks
            for i in range(5 * step, backlog, 3000):
                yield sig[i:min(i + 3000, backlog)]
            for k, i in enumerate(range(backlog, len(sig), 2048)):
                # live, in bursts of two chunks (a producer's wake-up, a transport's message): the burst takes a while to exist,
                # its second chunk is there at once -- a block that the SECOND chunk completes still goes out at once
                if k % 2 == 0:
                    time.sleep(2e-3)
                    asked.append((i, len(out)))       # what had been delivered when the loop came back for this burst
                yield sig[i:i + 2048]
        rb, _ = b.run_stream(source(), sink=out.append)
        assert len(out) == nblocks == len(ra)
        for x, y in zip(ra, out):
            assert x['count'] == y['count'] and _same(x['data'], y['data']) and _same(x['trust'], y['trust']) and _same(x['SNR'], y['SNR'])
        assert sum(sizes) == nblocks and max(sizes) >= 5, sizes                 # the backlog went through in batches
        live_sizes = sizes[-(nblocks - 15):]
        assert all(n == 1 for n in live_sizes), sizes                           # the live part block by block
        # whenever the loop asked for a live chunk, every block complete by then had been delivered (the first live requests
        # aside: the backlog's last batch is collected when the source turns out to be dry)
        for i, n in asked[2:]:
            assert n == i // step, (i, n)
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize('mod,pname,bs,D,snr', [('GMSK', 'bench_GMSK', 15, 32, 8.0), ('FSK', 'bench_FSK', 15, 16, 12.0),
                                                 ('GFSK', 'bench_GFSK', 14, 16, 12.0), ('BPSK', 'bench_BPSK', 15, 24, 10.0)])
def test_device_stream_stages_equal_the_host_stages(mod, pname, bs, D, snr):
    """The batched loop with A12 / A13 / A14 on the device (the default) against the same loop with them on the host
    ("HIP": {"stream_stages": false}) and against the one-block loop: bits, trust, centres, packets, alignment state and the
    decoder's overlap buffer -- for the bit-LUT modulations and for BPSK's NRZ-S decode; noiseless zero padding between the
    packets makes some blocks irregular (symbol index -1: no sample above zero), which go through the host code."""
    N, ov = 1 << bs, 1 << 10
    step = N - ov
    conf = cfg.bench_config(pname, blockSize=bs, doppCarrierSteps=D)
    p = loadProtocol(pname)(conf=conf)
    B, nblocks = 4, 19
    sig = sg.s1_stream(nblocks, N, ov, mod, snr_db=snr, seed=9)[ov:]
    confB, confH = copy.deepcopy(conf), copy.deepcopy(conf)
    confB['GPU']['UHF'].setdefault('HIP', {})['blocks_per_call'] = B
    confH['GPU']['UHF'].setdefault('HIP', {}).update(blocks_per_call=B, stream_stages=False)
    runs = [DemodulatorRunner(c, p, 'UHF-H') for c in (_one(conf), confB, confH)]
    decs = [Decoder(conf, p) for _ in runs]
    try:
        outs = [r.run_stream((sig[i:i + 7000] for i in range(0, len(sig), 7000)), decoder=d) for r, d in zip(runs, decs)]
        (r0, p0), (r1, p1), (r2, p2) = outs
        assert len(r0) == len(r1) == len(r2) == nblocks
        for x, y, z in zip(r0, r1, r2):
            for k in ('doppler', 'doppler_std', 'SNR', 'spSymEst', 'numSyncSig'):
                assert _same(x[k], y[k]) and _same(x[k], z[k]), (x['count'], k)
            assert _same(x['data'], y['data']) and _same(x['trust'], y['trust']) and _same(x['data'], z['data']), x['count']
            assert y['data'].dtype == np.uint8 and y['trust'].dtype == np.uint8
        assert len(p0) == len(p1) == len(p2) >= 1 and all(_same(u.bits, v.bits) for u, v in zip(p0, p1))
        for r in runs[1:]:
            assert _same(runs[0].demod.poswinP, r.demod.poswinP) and _same(runs[0].demod.posSymEnd, r.demod.posSymEnd)
        assert _same(decs[0].bitsOverlapBuf, decs[1].bitsOverlapBuf)
        assert runs[1].demod.stage_blocks >= nblocks // 2 and getattr(runs[2].demod, 'stage_blocks', 0) == 0, runs[1].demod.stage_blocks
        assert decs[1].ahead_blocks >= nblocks // 2
    finally:
        for r in runs:
            r.close()


def test_batches_fall_back_to_the_one_block_loop_where_they_do_not_apply():
    """A handle on the two-pass search path (filters without a short impulse response) or in the opt-in energy search mode runs
    one block per call whatever ``blocks_per_call`` says: same results, a warning."""
    bs, ov = 14, 1 << 10
    N = 1 << bs
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=16)
    confB = copy.deepcopy(conf)
    confB['GPU']['UHF'].setdefault('HIP', {}).update(blocks_per_call=4, search_path='twopass')
    conf['GPU']['UHF'].setdefault('HIP', {}).update(search_path='twopass')
    p = loadProtocol('bench_GMSK')(conf=conf)
    sig = sg.s1_stream(7, N, ov, 'GMSK', snr_db=10.0, seed=2)[ov:]
    a, b = DemodulatorRunner(_one(conf), p, 'UHF-H'), DemodulatorRunner(confB, p, 'UHF-H')
    try:
        ra, _ = a.run_stream([sig])
        rb, _ = b.run_stream([sig])
        assert len(ra) == len(rb) == 7 and all(_same(x['data'], y['data']) and _same(x['SNR'], y['SNR']) for x, y in zip(ra, rb))
        assert getattr(b.demod, 'stage_blocks', 0) == 0
    finally:
        a.close()
        b.close()


def test_batch_errors():
    bs = 15
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=8)
    p = loadProtocol('bench_GMSK')(conf=conf)
    d = UHF.Demodulator(conf, p, 'UHF-H')
    try:
        with pytest.raises(Exception):
            d.beginBlocks(0, 2)                 # no window yet
        d.blockWindows(3)
        with pytest.raises(Exception):
            d.beginBlocks(0, 4)                 # more blocks than the window holds
        d.beginBlocks(0, 2)
        with pytest.raises(Exception):
            d.beginBlocks(0, 2)                 # the slot is in flight
        with pytest.raises(Exception):
            d.bank.end_block(0)                 # a batch is not a block
        assert len(d.endBlocks(0)) == 2
        d.bank.set_search_path('twopass')
        with pytest.raises(ValueError):
            d.beginBlocks(0, 2)                 # batches run on the segment path
    finally:
        d.close()


def test_a_failing_source_or_sink_leaves_the_runner_usable():
    """The chunk source raises in the middle of a stream, then the sink does: the error reaches the caller, nothing stays in
    flight (copies queued for the copy thread included), and the same runner then processes a stream like a fresh one."""
    bs, ov, B = 14, 1 << 10, 4
    N = 1 << bs
    step = N - ov
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=16)
    conf['GPU']['UHF'].setdefault('HIP', {})['blocks_per_call'] = B
    p = loadProtocol('bench_GMSK')(conf=conf)
    sig = sg.s1_stream(30, N, ov, 'GMSK', snr_db=12.0, seed=2)[ov:]
    sig.flags.writeable = False
    run, fresh = DemodulatorRunner(conf, p, 'UHF-H'), DemodulatorRunner(conf, p, 'UHF-H')
    try:
        def failing_source():
            for i in range(0, len(sig), 5000):
                if i > 11 * step:
                    raise OSError('transport gone')
                yield sig[i:i + 5000]
        with pytest.raises(OSError, match='transport gone'):
            run.run_stream(failing_source())
        assert not run._copier._held and not run.demod.bank._flying

        seen = []

        def failing_sink(d):
            seen.append(d['count'])
            if len(seen) == 6:
                raise KeyError('consumer gone')
        with pytest.raises(KeyError):
            run.run_stream((sig[i:i + 5000] for i in range(0, len(sig), 5000)), sink=failing_sink)
        assert not run._copier._held and not run.demod.bank._flying
        # a clean stream afterwards: block by block what a fresh runner gives (the first block's alignment aside: it is made
        # against whatever block came last)
        ra, _ = run.run_stream([sig])
        rb, _ = fresh.run_stream([sig])
        assert len(ra) == len(rb) == 30
        for x, y in zip(ra[2:], rb[2:]):
            assert _same(x['data'], y['data']) and _same(x['doppler'], y['doppler']) and _same(x['SNR'], y['SNR']), (x['count'], y['count'])
    finally:
        run.close()
        fresh.close()


def test_c_entry_points_of_the_batch_path_refuse_bad_arguments():
    """mfb_window_buffer / mfb_receive_blocks_* / mfb_set_stream_stages / mfb_stream_seed called straight through ctypes with
    arguments outside their contract: a status code every time (MFB_ERR_ARG / MFB_ERR_STATE), never a crash, and the handle works
    afterwards."""
    import ctypes as C
    from pycusdr_amd import _lib
    bs = 13
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=8)
    p = loadProtocol('bench_GMSK')(conf=conf)
    d = UHF.Demodulator(conf, p, 'UHF-H')
    try:
        lib, h, bank = _lib.load(), d.bank._h, d.bank
        N, ov = 1 << bs, 1 << 10
        ptr = C.POINTER(C.c_float)()
        ARG, STATE = _lib.MFB_ERR_ARG, _lib.MFB_ERR_STATE
        for which, nb, stride in ((2, 2, N - ov), (-1, 2, N - ov), (0, 0, N - ov), (0, 2000, N - ov), (0, 2, 0), (0, 2, N + 1)):
            assert lib.mfb_window_buffer(h, which, nb, stride, C.byref(ptr)) == ARG, (which, nb, stride)
        assert lib.mfb_window_buffer(h, 0, 2, N - ov, None) == ARG and lib.mfb_window_buffer(None, 0, 2, N - ov, C.byref(ptr)) == ARG
        P = bank._block_params(d.codeRateAndPhaseOffsetLow if hasattr(d, 'codeRateAndPhaseOffsetLow') else 400, 200, 8, 0, 5, None, 'window', None)
        assert lib.mfb_receive_blocks_begin(h, C.byref(P), 2, 0) == STATE          # no window yet
        w = d.blockWindows(3)
        w[0][:] = sg.s1_stream(3, N, ov, 'GMSK', snr_db=12.0, seed=1)[:len(w[0])]
        for nb, slot in ((0, 0), (-3, 0), (2, 2), (2, -1)):
            assert lib.mfb_receive_blocks_begin(h, C.byref(P), nb, slot) == ARG, (nb, slot)
        assert lib.mfb_receive_blocks_begin(h, None, 2, 0) == ARG
        assert lib.mfb_receive_blocks_begin(h, C.byref(P), 4, 0) == STATE          # more blocks than the window holds
        Q = bank._block_params(400, 200, 8, 0, 5, None, 'window', None, block_stride=N - ov - 1)
        assert lib.mfb_receive_blocks_begin(h, C.byref(Q), 2, 0) == ARG            # not the window's stride
        Q = bank._block_params(400, 200, 8, 0, 5, None, 'device', None, block_stride=N - ov)
        assert lib.mfb_receive_blocks_begin(h, C.byref(Q), 2, 0) == ARG            # device input without a pointer
        lay = _lib.RecordLayout()
        buf = np.empty(1 << 20, np.uint8)
        assert lib.mfb_receive_blocks_end_record(h, 0, buf.ctypes.data, buf.size, C.byref(lay)) == STATE      # nothing in flight
        assert lib.mfb_receive_blocks_end_record(h, 3, buf.ctypes.data, buf.size, C.byref(lay)) == ARG
        assert lib.mfb_receive_blocks_end(h, 0, None, None, None, None, 0, None) != 0
        # stream stages
        S = _lib.StreamParams()
        assert lib.mfb_set_stream_stages(h, C.byref(S)) == ARG                    # all zero: no LUT mode
        lut = np.array([0, 1] * 4, np.uint8)
        S.overlap_samples, S.overlap_offset, S.match_threshold, S.error_threshold = ov, 20, 10, 1000
        S.lut_mode, S.lut_rows, S.lut = 1, 8, lut.ctypes.data
        for field, bad in (('lut_mode', 3), ('lut_rows', 0), ('lut_rows', 257), ('overlap_samples', 1), ('overlap_samples', N),
                           ('overlap_offset', 0), ('overlap_offset', 32), ('num_templates', 3), ('num_templates', -1)):
            keep = getattr(S, field)
            setattr(S, field, bad)
            assert lib.mfb_set_stream_stages(h, C.byref(S)) == ARG, (field, bad)
            setattr(S, field, keep)
        S.num_templates = 1                                                       # templates announced, none given
        assert lib.mfb_set_stream_stages(h, C.byref(S)) == ARG
        S.num_templates = 0
        assert lib.mfb_stream_seed(h, None, 0, None, 0, None, 0) == STATE          # stages are off
        assert lib.mfb_set_stream_stages(h, C.byref(S)) == 0
        z = np.zeros(5000, np.uint8)
        for npost, nend, nring in ((-1, 0, 0), (513, 0, 0), (0, 33, 0), (0, 0, 4097)):
            assert lib.mfb_stream_seed(h, z.ctypes.data, npost, z.ctypes.data, nend, z.ctypes.data, nring) == ARG, (npost, nend, nring)
        assert lib.mfb_stream_seed(h, None, 4, None, 0, None, 0) == ARG            # a length without an array
        assert lib.mfb_stream_seed(h, None, 0, None, 0, None, 0) == 0
        assert lib.mfb_set_stream_stages(h, None) == 0                            # off again
        # the entry points of round 6 (ABI 9): the scores of a batch's blocks, the two-stream arrangement, the search's description
        sc = np.empty(d.num_dopplers * 8 + 64, np.float32)
        fs_, bpf = C.c_int(-1), C.c_int(-1)
        assert lib.mfb_get_batch_scores(h, 0, sc.ctypes.data) == STATE           # no batch yet
        assert lib.mfb_get_batch_scores(h, -1, sc.ctypes.data) == ARG
        assert lib.mfb_get_batch_scores(h, 0, None) == ARG and lib.mfb_get_batch_scores(None, 0, sc.ctypes.data) == ARG
        for on in (-1, 2):
            assert lib.mfb_set_batch_overlap(h, on) == ARG
        assert lib.mfb_set_batch_overlap(None, 1) == ARG
        assert lib.mfb_get_search_info(h, None, C.byref(bpf)) == ARG and lib.mfb_get_search_info(h, C.byref(fs_), None) == ARG
        assert lib.mfb_get_search_info(None, C.byref(fs_), C.byref(bpf)) == ARG
        assert lib.mfb_get_search_info(h, C.byref(fs_), C.byref(bpf)) == 0 and fs_.value in (0, 1) and bpf.value >= 1
        # the handle still works
        d.beginBlocks(0, 3)
        assert lib.mfb_set_batch_overlap(h, 1) == STATE                           # a batch is in flight
        got = d.endBlocks(0)
        assert len(got) == 3 and all(len(r['symbols']) > 200 for _, r in got)
        assert lib.mfb_get_batch_scores(h, 3, sc.ctypes.data) == STATE           # the batch had three blocks
        assert lib.mfb_get_batch_scores(h, 2, sc.ctypes.data) == 0
        # ... and on two streams, switched while nothing is in flight: the same three blocks
        assert lib.mfb_set_batch_overlap(h, 1) == 0
        d.beginBlocks(0, 3)
        again = d.endBlocks(0)
        assert all(_same(a[1]['symbols'], b[1]['symbols']) for a, b in zip(got, again))
        assert lib.mfb_set_batch_overlap(h, 0) == 0
    finally:
        d.close()


def test_cc11xx_stream_with_blocks_per_call():
    """The production protocol (config/CC11xx.json: FSK-2 at 128 samples per symbol, 384-tap filters -> 2048-point segments, IF offset,
    numBitsOverlap 2048, a 64-tap header mask and a 32-tap sync flag, FIXED packets of 2136 bits) through the batched loop with the
    stream stages on the device, against the one-block loop: result dicts, bits, trust, packets (with their CRC flags), alignment state."""
    from pycusdr_amd.protocol.CC11xx import frame_bits
    bs, sps, B = 17, 128, 3
    N, ov = 1 << bs, 1 << 10
    step = N - ov
    conf = cfg.cc11xx_config(blockSize=bs, doppCarrierSteps=48, samplesPerSym=sps)
    p = loadProtocol('CC11xx')(conf=conf)
    rs = np.random.RandomState(4)
    fs = 7416 * sps
    nblocks = 10
    frames = [frame_bits(rs.randint(0, 256, 200).astype(np.uint8), preamble=(0xAA,) * 10) for _ in range(8)]
    bits = np.concatenate([np.concatenate((f, rs.randint(0, 2, 300).astype(f.dtype))) for f in frames])
    sig = sg.modulateFSK(bits, sps)
    sig = np.tile(sig, -(-(nblocks * step) // len(sig)))[:nblocks * step]
    sig = sg.awgn(sig * np.exp(2j * np.pi * 148320 / fs * np.arange(len(sig))), 12.0, rng=np.random.RandomState(2)).astype(np.complex64)
    confB = copy.deepcopy(conf)
    confB['GPU']['UHF'].setdefault('HIP', {})['blocks_per_call'] = B
    a, b = DemodulatorRunner(_one(conf), p, 'UHF-H'), DemodulatorRunner(confB, p, 'UHF-H')
    da, db = Decoder(conf, p), Decoder(conf, p)
    try:
        ra, pa = a.run_stream((sig[i:i + 30000] for i in range(0, len(sig), 30000)), decoder=da)
        rb, pb = b.run_stream((sig[i:i + 30000] for i in range(0, len(sig), 30000)), decoder=db)
        assert len(ra) == len(rb) == nblocks
        for x, y in zip(ra, rb):
            for k in ('doppler', 'doppler_std', 'SNR', 'spSymEst', 'numSyncSig'):
                assert _same(x[k], y[k]), (x['count'], k)
            assert _same(x['data'], y['data']) and _same(x['trust'], y['trust']), x['count']
        assert len(pa) == len(pb) >= 3
        for u, v in zip(pa, pb):
            assert _same(u.bits, v.bits) and u.frameStartIdx == v.frameStartIdx
        assert _same(a.demod.poswinP, b.demod.poswinP) and _same(da.bitsOverlapBuf, db.bitsOverlapBuf)
        assert b.demod.stage_blocks >= nblocks - 2 and db.ahead_blocks >= nblocks - 2
    finally:
        a.close()
        b.close()
