"""Protocol plugin API consumed by the Demodulator and the Decoder.

Mirrors the surface of the reference's ProtocolBase (protocol/protocolBase.py:27-165) that the hot
path touches: get_filter / get_symbolLUT2 on the demodulator side, get_mask / get_syncFlag,
tolerances, overlap and packet-end attributes on the decoder side.  Packet parsing (CRC,
de-whitening) is out of scope of this path; ``Packet`` keeps bits and bookkeeping only.
"""
import logging
from enum import Enum

import numpy as np

log = logging.getLogger('pycusdr_amd.protocol')

DEFAULT_NO_SYNC_FLAGS = 2


class PacketEndDetect(Enum):
    FLAGS = 0
    FIXED = 1
    IN_DATA = 2


class PacketLenEndianness(Enum):
    LITTLE = True
    BIG = False


def bit_patterns(maskLen):
    """All 2**maskLen bit patterns, MSB first, as float rows (reference `_get_xcorrMasks`,
    protocol/CC11xx.py:80-86, bench_base.py:46-53)."""
    idx = np.arange(1 << maskLen)[:, None]
    sh = np.arange(maskLen - 1, -1, -1)[None, :]
    return ((idx >> sh) & 1).astype(np.float64)


def bank_from_templates(templates, Nfft):
    """conj(fft(template, Nfft)) per row as complex64 -- the layout the matched-filter bank takes."""
    out = np.empty((len(templates), Nfft), dtype=np.complex64)
    for i, t in enumerate(templates):
        out[i] = np.conj(np.fft.fft(t, Nfft)).astype(np.complex64)
    return out.shape[0], out


class ProtocolBase:
    name = 'ProtocolBase'
    numBitsOverlap = 2 * 513
    packetEndDetectMode = PacketEndDetect.FLAGS
    packetLen = None
    packetEndLenField = None
    packetEndLenFieldNumBytes = None
    packet_sizes = []

    def __init__(self, **args):
        pass

    # -- demodulator side ---------------------------------------------------------------------
    def _get_xcorrMasks(self, maskLen):
        return bit_patterns(maskLen)

    def get_filter(self, Nfft, spSym=None, maskSize=0):
        """-> (M, complex64[M, Nfft]): conjugated spectra of the 2**maskSize symbol templates."""
        raise NotImplementedError('Sub class needs to implement this method')

    def get_symbolLUT2(self, maskLen):
        """-> (bitLUT | None, symbolLUT): centre-bit lookup per filter index."""
        raise NotImplementedError('Sub class needs to implement this method')

    # -- decoder side ---------------------------------------------------------------------------
    def get_mask(self):
        raise NotImplementedError

    def get_syncFlag(self):
        raise NotImplementedError

    def decoderPreprocessor(self, signal, **args):
        return signal

    def decoderPostprocessor(self, packet, **args):
        return packet

    def packetDataProcessor(self, packet=None):
        return None

    def packetEndLenDecoder(self, bits, **args):
        return 0

    def Packet(self, *args, **kwargs):
        return Packet(self, *args, **kwargs)

    def __repr__(self):
        return f'<protocol {self.name}>'


class Packet:
    """Bits of one received frame plus where it was found."""

    def __init__(self, protocol, bits, frameStartIdx=0, maskBitErrors=0, frameSplitIdx=0, **kwargs):
        self.protocol = protocol
        self.bits = np.asarray(bits)
        self.frameStartIdx = frameStartIdx
        self.maskBitErrors = maskBitErrors
        self.frameSplitIdx = frameSplitIdx

    @property
    def bitsRaw(self):
        return self.bits

    def getBinaryRawData(self):
        return self.bits

    def getBinaryData(self):
        nbytes = len(self.bits) // 8
        data = np.dot(self.bits[:nbytes * 8].reshape(nbytes, 8), 2 ** np.arange(0, 8, 1)).astype(np.uint8)
        return data, 0, self.bits
