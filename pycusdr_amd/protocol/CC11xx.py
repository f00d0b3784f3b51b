"""CC11xx (TI sub-GHz transceiver framing) receive plugin: 2-FSK (or Gaussian 2-FSK) filters,
centre-bit LUT, preamble+sync template, and the packet class with the length-byte cut, PN9
de-whitening and the CRC flag (reference protocol/CC11xx.py:42-146, 216-300).

Two things the reference fixes at import time are selectable here:
  * modulation (reference ``modIDX``, CC11xx.py:30-32,42): ``CC11xx`` is FSK-2, ``CC11xx_GFSK2`` the
    GFSK-2 flavour (``loadProtocol('CC11xx_GFSK2')``);
  * where the CRC is looked for (``CRC_CHECK``): ``'reference'`` (default) reproduces the reference's
    ``getBinaryData`` bit for bit -- the flag compares crc16([len | packetLen bytes]) with the two RAW
    bytes that follow them; ``'framer'`` checks the CRC where the reference's own TX framer puts it
    (inside the length-counted bytes, modulator/encoders/CC11xx.py:82-96).
"""
import numpy as np

from .FSK2_base import FSK2
from .GFSK2_base import GFSK2
from .protocolBase import Packet, PacketEndDetect, PacketLenEndianness, bit_patterns

DEFAULT_SYNC = [0xAB, 0x35, 0xAB, 0x35]
DEFAULT_PREAMBLE = [0xAA]
DEFAULT_NUM_PREAMBLE = 4


def _hex_list(v):
    return [int(k, 16) if isinstance(k, str) else int(k) for k in v]


def _bytes_to_bits(byte_list):
    return np.unpackbits(np.asarray(byte_list, dtype=np.uint8)).astype(np.float64)


def pn9_bytes(n):
    """PN9 whitening bytes of the CC11xx (x^9 + x^5 + 1, all-ones start, one byte per 8 clocks):
    0xFF 0xE1 0x1D 0x9A ...  Same sequence as the reference's lib/shift_registers.PN9 (:76-91);
    pinned by fixture G8."""
    reg = [1] * 9
    out = np.empty(n, dtype=np.int64)
    for i in range(n):
        out[i] = sum(reg[k] << k for k in range(8))
        for _ in range(8):
            reg = reg[1:] + [reg[0] ^ reg[5]]
    return out


def pn9_first_byte():
    """First whitening byte of the PN9 sequence (all-ones start): 0xFF."""
    return 0xFF


def crc16_cc11xx(data):
    """CRC-16 of the CC11xx: polynomial 0x8005, initial value 0xFFFF, MSB first, no final XOR
    (what the reference builds with crcmod.mkCrcFun(0x18005, rev=False, initCrc=0xFFFF, xorOut=0),
    CC11xx.py:255; catalogue name CRC-16/CMS, check value 0xAEE7 for b"123456789")."""
    crc = 0xFFFF
    for byte in bytes(bytearray(int(b) & 0xFF for b in data)):
        crc ^= byte << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x8005) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    return crc


def frame_bits(payload, preamble=(0xAA,) * 4, sync=(0xD6, 0xBA, 0xD6, 0xBA), whiten=True):
    """Bits of one CC11xx frame as the reference's TX framer builds it (modulator/encoders/CC11xx.py
    :64-118): preamble | sync | whiten([len = n+2 | payload | CRC low byte | CRC high byte]), MSB first.
    Test stimulus for the receive path; the transmit chain itself is out of scope."""
    payload = np.asarray(payload, dtype=np.uint8)
    body = np.r_[len(payload) + 2, payload].astype(np.uint8)
    crc = crc16_cc11xx(body)
    body = np.r_[body, crc & 0xFF, crc >> 8].astype(np.uint8)
    if whiten:
        body = np.bitwise_xor(body, pn9_bytes(len(body)).astype(np.uint8))
    return np.unpackbits(np.r_[np.asarray(preamble, np.uint8), np.asarray(sync, np.uint8), body])


class _CC11xxFraming:
    """Everything of the plugin but the filter generator (mixed into an FSK2 / GFSK2 base)."""
    CRC_CHECK = 'reference'      # or 'framer'
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


class CC11xx(_CC11xxFraming, FSK2):
    name = 'CC11xx FSK-2'


class CC11xx_GFSK2(_CC11xxFraming, GFSK2):
    name = 'CC11xx GFSK-2'


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
        self._pre = DEFAULT_NUM_PREAMBLE + self.maskLen + self.pLen      # bytes before the length-counted part

    def getBinaryData(self):
        """(packetLen bytes after the length byte, de-whitened; CRC flag; the same bytes) -- the return
        shape of the reference (CC11xx.py:274-300).

        ``CRC_CHECK == 'reference'``: the flag is the reference's, bit for bit: crc16 over
        [len | the packetLen de-whitened bytes] compared with the last two bytes of the cut frame taken RAW
        (still whitened), low byte first; True means mismatch.  For frames of the reference's own TX
        framer (length byte counts the CRC) that flag is set on every good frame.
        ``CRC_CHECK == 'framer'``: the CRC is the last two of the packetLen bytes, computed over
        [len | payload]; True means CRC error."""
        n = int(np.uint8(self.packetLen))
        body = np.asarray(self.bits[self._pre * 8:(self._pre + n) * 8]).astype(np.int64)
        data = np.dot(body.reshape(-1, 8), 2 ** np.arange(7, -1, -1)).astype(np.uint8)
        if self.protocol.deWhiten:
            k = min(len(data), int(self.packetLen))
            data[:k] = np.bitwise_xor(data[:k], pn9_bytes(k + 1)[1:].astype(np.uint8))
        if getattr(self.protocol, 'CRC_CHECK', 'reference') == 'framer':
            if len(data) < 2 or len(data) < n:
                return data, True, data
            crc_rx = int(data[-2]) | (int(data[-1]) << 8)
            return data, crc16_cc11xx(np.r_[n, data[:-2]]) != crc_rx, data
        tail = np.asarray(self.bits[-self.CRClen * 8:]).astype(np.int64).reshape(self.CRClen, 8)
        crc_bytes = np.dot(tail, 2 ** np.arange(7, -1, -1))
        crc_rx = int(crc_bytes[0]) + 256 * int(crc_bytes[1])
        return data, crc_rx != crc16_cc11xx(np.r_[n, data].astype(np.uint8)), data
