"""Decoder.findFrames host logic against KATs recorded from the reference's Decoder (fixtures G4).
The correlator is injected (oracle, CPU) so the packet state machine is pinned without a GPU; the
GPU correlator itself is covered by tests/test_gpu_decoder.py."""
import numpy as np
import pytest

from oracle import mfbank_oracle as orc
from pycusdr_amd import config as cfg
from pycusdr_amd.decoder import Decoder
from pycusdr_amd.protocol import PacketEndDetect, loadProtocol
from pycusdr_amd.protocol.protocolBase import ProtocolBase

CASES = [('bench', 'bench_GMSK', cfg.bench_config()), ('CC11xx', 'CC11xx', cfg.cc11xx_config())]


def run_kat(goldens, key, pname, conf, correlator):
    p = loadProtocol(pname)(conf=conf)
    d = Decoder({}, p, correlator=correlator)
    stream = goldens[f'g4/{key}/stream'].astype(np.float64)
    cuts = goldens[f'g4/{key}/cuts']
    for ci in range(3):
        packets, bits, nsync = d.findFrames(stream[cuts[ci]:cuts[ci + 1]], 0)
        assert nsync == int(goldens[f'g4/{key}/call{ci}/numSyncSig'])
        assert len(packets) == int(goldens[f'g4/{key}/call{ci}/npackets'])
        assert np.array_equal(np.asarray(d.bitsOverlapBuf).astype(np.uint8), goldens[f'g4/{key}/call{ci}/overlapBuf_after'])
        assert np.array_equal(bits, stream[cuts[ci]:cuts[ci + 1]])
        for i, q in enumerate(packets):
            assert np.array_equal(np.asarray(q.bits).astype(np.uint8), goldens[f'g4/{key}/call{ci}/p{i}/bits'])
            if key == 'bench':
                assert q.frameStartIdx == int(goldens[f'g4/{key}/call{ci}/p{i}/start'])
                assert q.maskBitErrors == float(goldens[f'g4/{key}/call{ci}/p{i}/maskBitErrors'])


@pytest.mark.parametrize('key,pname,conf', CASES)
def test_findframes_matches_reference_kats(goldens, key, pname, conf):
    run_kat(goldens, key, pname, conf, orc.sync_correlate)


def test_bench_packet_bit_error_count():
    p = loadProtocol('bench_GMSK')(conf=cfg.bench_config())
    good = np.random.RandomState(123).randint(0, 2, 10000)
    pk = p.Packet(good, 0, 0)
    assert pk.checkPacketData() == 0
    bad = good.copy()
    bad[[5, 77, 9000]] ^= 1
    assert p.Packet(bad, 0, 0).checkPacketData() == 3
    assert p.Packet(good[:500], 0, 0).checkPacketData() == -0.1


class _FlagsProto(ProtocolBase):
    """Minimal FLAGS-mode plugin (the reference ships none that the tests could record): header =
    16 known bits, sync flag = 0x7e twice."""
    name = 'flags-test'
    packetEndDetectMode = PacketEndDetect.FLAGS
    numBitsOverlap = 64
    headerTol = 0
    syncSigTol = 0
    HDR = np.array([1, 1, 1, 0, 0, 0, 1, 0, 0, 1, 0, 1, 1, 0, 1, 1])
    FLAG = np.array([0, 1, 1, 1, 1, 1, 1, 0] * 2)

    def get_mask(self):
        self.numOnesHeader = self.HDR.sum()
        return np.flipud(self.HDR * 2 - 1)

    def get_syncFlag(self):
        self.numOnesSyncSig = self.FLAG.sum()
        return np.flipud(self.FLAG * 2 - 1)


def test_flags_mode_within_one_block_and_across_blocks():
    p = _FlagsProto()
    rs = np.random.RandomState(4)
    body = np.zeros(300, dtype=np.float64)      # all-zero payload cannot imitate header or flag
    # a flag on either side, as in a real flag-delimited stream: the reference can only end a frame
    # on the 2nd or later sync hit of a block (argmax == 0 means "none", decoder.py:211-213)
    frame = np.concatenate((p.FLAG, p.HDR, body, p.FLAG))
    stream = np.concatenate((np.zeros(100), frame, np.zeros(400)))
    d = Decoder({}, p, correlator=orc.sync_correlate)
    packets, _, nsync = d.findFrames(stream, 1000)
    assert nsync >= 1 and len(packets) == 1
    q = packets[0]
    assert q.frameStartIdx == 1000 + 100 + 16 + p.numBitsOverlap
    assert np.array_equal(q.bits[:16], p.HDR) and len(q.bits) >= 128
    # same frame cut in the middle: header in call 1, flag in call 2 -> one packet, split recorded
    d = Decoder({}, p, correlator=orc.sync_correlate)
    cut = 100 + 16 + 150
    p1, _, _ = d.findFrames(stream[:cut], 0)
    assert p1 == [] and d.headerFrameStartIdx is not None
    p2, _, _ = d.findFrames(stream[cut:], 0)
    assert len(p2) == 1 and np.array_equal(p2[0].bits[:16], p.HDR) and p2[0].frameSplitIdx > 0
    assert d.headerFrameStartIdx is None
    del rs


def _packets_equal(pa, pb):
    assert len(pa) == len(pb)
    for x, y in zip(pa, pb):
        assert np.array_equal(np.asarray(x.bits), np.asarray(y.bits))
        assert x.frameStartIdx == y.frameStartIdx and x.maskBitErrors == y.maskBitErrors
        assert getattr(x, 'frameSplitIdx', None) == getattr(y, 'frameSplitIdx', None)


def _bench_stream(rs, nblocks, blk, flips):
    """Back-to-back seed-123 bench packets (10 000 bits: their first 128 bits are the header, a packet spans several blocks, so
    the FIXED-mode stash DEC:254-263 is live in most calls), noise bits in between, a few bit errors in the headers."""
    pkt = np.random.RandomState(123).randint(0, 2, 10000)
    parts = []
    while sum(len(x) for x in parts) < nblocks * (blk + 2):
        q = pkt.copy()
        q[rs.randint(0, 128, flips)] ^= 1
        parts += [rs.randint(0, 2, rs.randint(0, 700)), q]
    return np.concatenate(parts).astype(np.uint8)


@pytest.mark.parametrize('with_ahead', [False, True])
@pytest.mark.parametrize('blk,B,flips', [(1950, 8, 0), (1950, 5, 20), (2500, 3, 30), (700, 16, 26), (9000, 2, 0), (130, 16, 10)])
def test_findframes_batch_equals_call_by_call_fixed(blk, B, flips, with_ahead):
    """findFrames_batch (searches run ahead on the default windows, stashed windows put together from the previous call's hits)
    against findFrames call by call: packets, returned bits, sync counts and the overlap buffer, block by block."""
    p = loadProtocol('bench_GMSK')(conf=cfg.bench_config())
    rs = np.random.RandomState(blk + B)
    nblocks = 4 * B + 3
    stream = _bench_stream(rs, nblocks, blk, flips)
    cuts = np.cumsum([0] + [blk + int(rs.randint(-1, 2)) for _ in range(nblocks)])
    blocks = [stream[cuts[i]:cuts[i + 1]] for i in range(nblocks)]
    a, b = Decoder({}, p, correlator=orc.sync_correlate), Decoder({}, p, correlator=orc.sync_correlate)
    direct = [0]
    inner = b.hits

    def counting(bits, template, threshold):
        direct[0] += 1
        return inner(bits, template, threshold)
    want = [a.findFrames(x, 7) for x in blocks]
    got = []
    # with_ahead: the hits of every block's stream without a stash come with the block, as the batched block path delivers them
    # (here: the oracle's correlator on the bit sequence), some of them missing; no search but the new-stash edges is left
    nOv = p.numBitsOverlap
    seq = np.concatenate((np.zeros(nOv), stream[:cuts[-1]]))
    thr = (p.numOnesHeader - p.headerTol, p.numOnesSyncSig - p.syncSigTol)
    missing_total = 0
    for i in range(0, nblocks, B):
        group = blocks[i:i + B]
        ahead = edges = None
        if with_ahead:
            ahead, edges = [], []
            Ts = (len(b.mask), len(b.syncSig))
            for j in range(i, min(i + B, nblocks)):
                w = seq[cuts[j]:cuts[j + 1] + nOv]
                ahead.append(None if (j % 7 == 3) else tuple(inner(w, t, h) for t, h in zip((b.mask, b.syncSig), thr)))
                cands = []
                for idx in ([] if ahead[-1] is None else ahead[-1][0][0][:4]):
                    a_rel = int(idx) - Ts[0] + 1 - 20
                    if cuts[j] + a_rel >= 0:
                        lead = seq[cuts[j] + a_rel: cuts[j] + a_rel + max(Ts) - 1]
                        hh = []
                        for t, h, T in zip((b.mask, b.syncSig), thr, Ts):
                            ei, es = inner(lead, t, h)
                            hh.append((ei[ei < T - 1], es[ei < T - 1]))
                        cands.append((a_rel, hh[0], hh[1]))
                edges.append(cands if ahead[-1] is not None else None)
        b.hits = counting
        before = direct[0]
        got += b.findFrames_batch(group, 7, ahead=ahead, edges=edges)
        if with_ahead:
            # no search is left but those of the blocks whose hits did not come with them, and of a stash made behind such a block
            missing_total += sum(1 for x in ahead if x is None)
            assert direct[0] <= 4 * missing_total + 2, (i, direct[0], missing_total)
        # searches: two per block for the run-ahead windows, plus two per NEW stash (once per packet, not once per block)
        assert direct[0] - before <= 2 * len(group) + 2 * 3, (i, direct[0] - before)
    found = 0
    for (pw, bw, nw), (pg, bg, ng) in zip(want, got):
        _packets_equal(pw, pg)
        assert np.array_equal(bw, bg) and nw == ng
        found += len(pw)
    assert found >= 1 or blk * nblocks < 12000
    assert np.array_equal(np.asarray(a.bitsOverlapBuf), np.asarray(b.bitsOverlapBuf))


def test_findframes_batch_equals_call_by_call_flags_and_cc11xx():
    rs = np.random.RandomState(8)
    p = _FlagsProto()
    frame = np.concatenate((p.FLAG, p.HDR, np.zeros(300), p.FLAG))
    stream = np.concatenate([np.concatenate((rs.randint(0, 2, rs.randint(20, 200)) * 0, frame)) for _ in range(12)] + [np.zeros(300)])
    cuts = np.cumsum([0] + [int(rs.randint(150, 400)) for _ in range(200)])
    cuts = cuts[cuts < len(stream)]
    blocks = [stream[cuts[i]:cuts[i + 1]] for i in range(len(cuts) - 1)]
    a, b = Decoder({}, p, correlator=orc.sync_correlate), Decoder({}, p, correlator=orc.sync_correlate)
    want = [a.findFrames(x, 5) for x in blocks]
    got = []
    for i in range(0, len(blocks), 6):
        got += b.findFrames_batch(blocks[i:i + 6], 5)
    assert sum(len(w[0]) for w in want) >= 8
    for (pw, bw, nw), (pg, bg, ng) in zip(want, got):
        _packets_equal(pw, pg)
        assert nw == ng
    assert a.headerFrameStartIdx == b.headerFrameStartIdx
    # the reference's CC11xx KAT stream, cut as recorded, as one batch
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ref_goldens.npz'))
    g = {k.replace('__', '/'): z[k] for k in z.files}
    pc = loadProtocol('CC11xx')(conf=cfg.cc11xx_config())
    s2 = g['g4/CC11xx/stream'].astype(np.float64)
    c2 = g['g4/CC11xx/cuts']
    a, b = Decoder({}, pc, correlator=orc.sync_correlate), Decoder({}, pc, correlator=orc.sync_correlate)
    want = [a.findFrames(s2[c2[i]:c2[i + 1]], 0) for i in range(3)]
    got = b.findFrames_batch([s2[c2[i]:c2[i + 1]] for i in range(3)], 0)
    for (pw, bw, nw), (pg, bg, ng) in zip(want, got):
        _packets_equal(pw, pg)
        assert nw == ng
    assert np.array_equal(np.asarray(a.bitsOverlapBuf), np.asarray(b.bitsOverlapBuf))


def test_a_decoder_is_released_with_its_last_reference():
    """A decoder owns device-side state (the sync finder: a stream, templates, page-locked staging).  It must not sit in a
    reference cycle -- then that state would stay until the cycle collector next runs, and a process that creates receivers
    in a loop grows (tools/leak_probe.py: +0.1 MiB of host memory per create / stream / close cycle before this)."""
    import gc
    import weakref
    was = gc.isenabled()
    gc.disable()
    try:
        for correlator in (None, orc.correlate_convolve if hasattr(orc, 'correlate_convolve') else (lambda b, t: np.convolve(b, t))):
            conf = cfg.bench_config()
            d = Decoder(conf, loadProtocol('bench_GMSK')(conf=conf), correlator=correlator)
            if correlator is not None:          # with host work behind it: blocks through the packet state machine
                rs = np.random.RandomState(5)
                for _ in range(3):
                    d.findFrames(rs.randint(0, 2, 4000).astype(np.uint8), 0)
            r = weakref.ref(d)
            d.close()                            # explicit release is there too, and harmless without a finder
            del d
            assert r() is None
    finally:
        if was:
            gc.enable()
