"""CC11xx (TI sub-GHz transceiver framing) receive plugin: 2-FSK filters, centre-bit LUT,
preamble+sync template.  Mirrors the parts of the reference's protocol/CC11xx.py the hot path uses
(:42-146); the packet class keeps the length-byte cut (:254-266) but no CRC / PN9 de-whitening of
the payload (next scope row N2).
"""
import numpy as np

from .FSK2_base import FSK2
from .protocolBase import Packet, PacketEndDetect, PacketLenEndianness, bit_patterns

DEFAULT_SYNC = [0xAB, 0x35, 0xAB, 0x35]
DEFAULT_PREAMBLE = [0xAA]
DEFAULT_NUM_PREAMBLE = 4


def _hex_list(v):
    return [int(k, 16) if isinstance(k, str) else int(k) for k in v]


def _bytes_to_bits(byte_list):
    return np.unpackbits(np.asarray(byte_list, dtype=np.uint8)).astype(np.float64)


def pn9_first_byte():
    """First whitening byte of the PN9 sequence (all-ones start): 0xFF."""
    return 0xFF


class CC11xx(FSK2):
    name = 'CC11xx FSK-2'
    packetEndDetectMode = PacketEndDetect.FIXED
    packetLen = (256 + 9 + 2) * 8
    packetEndLenField = 9
    packetEndLenFieldNumBytes = 1
    packetEndLenEndianness = PacketLenEndianness.LITTLE
    deWhiten = True
    whiten = True
    SUM_ALL_MASKS_PYTHON = True
    numBitsOverlap = 2048

    numOnesSyncSig = 0
    numOnesHeader = 0
    syncSigTol = 2
    headerTol = 5

    def __init__(self, **args):
        cfg = args.get('conf', None)
        prot = cfg['Radios'].get('Protocol', None) if cfg else None
        if prot:
            self.rx_preamble = _hex_list(prot['rx_preamble'])
            self.rx_sync_seq = _hex_list(prot['rx_sync_seq'])
            self.tx_preamble = _hex_list(prot['tx_preamble'])
            self.tx_num_preambles = prot['tx_num_preambles']
            self.tx_sync_seq = _hex_list(prot['tx_sync_seq'])
        else:
            # reference defaults when no config is given (CC11xx.py:71-77): note that it leaves
            # rx_preamble unset there; we give it the default preamble so get_mask works
            self.rx_preamble = DEFAULT_PREAMBLE * DEFAULT_NUM_PREAMBLE
            self.rx_sync_seq = DEFAULT_SYNC * 4
            self.tx_preamble = DEFAULT_PREAMBLE
            self.tx_num_preambles = DEFAULT_NUM_PREAMBLE
            self.tx_sync_seq = DEFAULT_SYNC
        self.num_preamble_bytes = len(self.tx_preamble * self.tx_num_preambles)

    def get_symbolLUT2(self, maskLen):
        pats = bit_patterns(maskLen)
        bitLUT = pats[:, int(maskLen / 2)]
        half = np.arange(1 << (maskLen - 1))
        symLUT = np.stack((half * 2 + 1, half * 2), axis=1).astype(int)
        return bitLUT, np.concatenate((symLUT, symLUT), axis=0)

    def get_mask(self):
        bits = _bytes_to_bits(self.rx_preamble + self.rx_sync_seq)
        self.numOnesHeader = np.sum(bits)
        return np.flip(bits * 2 - 1, axis=0)

    def get_syncFlag(self):
        bits = _bytes_to_bits(self.rx_preamble)
        self.numOnesSyncSig = np.sum(bits > 0)
        return bits * 2 - 1

    def Packet(self, *args, **kwargs):
        return PacketCC11xx(self, *args, **kwargs)


class PacketCC11xx(Packet):
    """flags | sync | length byte | address | data | CRC -- cut to the length the (whitened)
    length byte announces (reference CC11xx.py:226-266)."""
    packetLenFieldIndex = 8
    maskLen = 4
    pLen = 1
    CRClen = 2

    def __init__(self, protocol, bits, frameStartIdx=0, maskBitErrors=0, frameSplitIdx=0, **kwargs):
        bits = np.asarray(bits)
        i0 = self.packetLenFieldIndex * 8
        raw_len = int(np.sum(bits[i0:i0 + 8] * 2 ** np.arange(7, -1, -1)))
        self.packetLen = (raw_len ^ pn9_first_byte()) if protocol.deWhiten else raw_len
        # the reference computes its overhead from a module constant fixed at import
        # (DEFAULT_NUM_PREAMBLE = 4 flag bytes): flags + sync + len + CRC
        overhead = DEFAULT_NUM_PREAMBLE + self.maskLen + self.pLen + self.CRClen
        super().__init__(protocol, bits[:int(self.packetLen + overhead) * 8], frameStartIdx, maskBitErrors, frameSplitIdx)
