"""Base of the benchmark protocols: fixed-length packets with a known seed-123 preamble
(reference protocol/benchmark/bench_base.py:21-205)."""
import logging

import numpy as np

from ..protocolBase import Packet, PacketEndDetect, ProtocolBase

log = logging.getLogger('pycusdr_amd.protocol')

MASKLEN = 16 * 8
FLAGLEN = 8 * 2
PACKETLEN = 1000
RAND_SEED = 123


def legacy_seeded_bits(seed, n):
    """np.random.seed(seed); np.random.randint(0, 2, n) without touching the global RNG state
    (RandomState reproduces the legacy global stream exactly)."""
    return np.random.RandomState(seed).randint(0, 2, n)


class Bench_base(ProtocolBase):
    name = 'bench_base_class'
    packetEndDetectMode = PacketEndDetect.FIXED
    packetLen = PACKETLEN
    numBitsOverlap = MASKLEN * 2

    numOnesSyncSig = 0
    numOnesHeader = 0
    syncSigTol = 1
    headerTol = 27

    def __init__(self, **args):
        conf = args.get('conf', None)
        main = conf.get('Main', {}) if conf else {}
        self.conf = conf
        self.packetLen = main.get('PacketLen', PACKETLEN)
        self.randSeed = main.get('RandSeed', RAND_SEED)

    def centre_bit_lut(self, maskLen):
        return self._get_xcorrMasks(maskLen)[:, int(maskLen / 2)]

    def get_mask(self):
        bits = legacy_seeded_bits(123, MASKLEN)
        self.numOnesHeader = np.sum(bits)
        return np.flipud(bits * 2 - 1)

    def get_syncFlag(self):
        bits = legacy_seeded_bits(123, FLAGLEN)
        self.numOnesSyncSig = np.sum(bits)
        return np.flipud(bits * 2 - 1)

    def Packet(self, *args, **kwargs):
        return Packet_bench(self, *args, **kwargs, packetLen=self.packetLen, randSeed=self.randSeed)


class Packet_bench(Packet):
    """Known-payload packet: bit errors are counted against the seeded sequence."""

    def __init__(self, protocol, bits, frameStartIdx, maskBitErrors, frameSplitIdx=0, packetLen=PACKETLEN,
                 randSeed=RAND_SEED):
        super().__init__(protocol, np.asarray(bits).astype(np.int8), frameStartIdx, maskBitErrors, frameSplitIdx)
        self.packetLen = packetLen
        self.randSeed = randSeed

    def checkPacketData(self):
        if len(self.bits) < self.packetLen:
            return -0.1
        expect = legacy_seeded_bits(self.randSeed, self.packetLen)
        return int(np.count_nonzero(self.bits[:self.packetLen] != expect))

    def getBinaryData(self):
        return self.bits, 0, self.bits
