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
