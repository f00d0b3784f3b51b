"""BPSK benchmark protocol: RRC-shaped +/-1 templates and the NRZ-S transition LUT
(reference protocol/benchmark/bench_BPSK.py:31-261)."""
import numpy as np

from ...lib.filters import rrcosfilter
from ..protocolBase import PacketEndDetect, bank_from_templates
from .bench_base import MASKLEN, PACKETLEN, Bench_base


def decodeNRZS(nrz):
    nrz = np.asarray(nrz)
    out = np.zeros(len(nrz), dtype=np.uint8)
    out[0] = nrz[0]
    out[1:] = (nrz[1:] == nrz[:-1]).astype(np.uint8)
    return out


def _nrzs_lut(maskLen):
    """symbolLUT[sym][0] = successor filters that decode to bit 1, [sym][1] = to bit 0 (NRZ-S: a 1
    keeps the line level, a 0 toggles it).

    Only the first 2**(maskLen-1) filters are listed: a filter and its bit-complement have the same
    |.|^2 and the strict '>' of the centre search keeps the lower index.  With the filters split
    into four equal groups G0..G3 by their two leading bits, rows in G0/G3 map to [G0, G1] and rows
    in G1/G2 to [G3, G2].  Generated here; pinned entry by entry against the reference's literal
    tables (bench_BPSK.py:84-197) by tests/test_protocols.py.
    """
    if maskLen not in (4, 5):
        raise Exception(f'bench_BPSK: Invalid mask length ({maskLen})')
    half = 1 << (maskLen - 1)
    q = half // 4
    groups = [np.arange(i * q, (i + 1) * q) for i in range(4)]
    lut = np.zeros((half, 2, q), dtype=int)
    for s in range(half):
        outer = (s // q) in (0, 3)
        lut[s, 0] = groups[0] if outer else groups[3]
        lut[s, 1] = groups[1] if outer else groups[2]
    return lut


class Bench_BPSK(Bench_base):
    name = 'bench_BPSK'
    packetEndDetectMode = PacketEndDetect.FIXED
    packetLen = PACKETLEN
    numBitsOverlap = MASKLEN * 2
    SUM_ALL_MASKS_PYTHON = True

    def get_filter(self, Nfft, spSym, maskSize):
        self.num_masks = int(2 ** (maskSize - 1))
        pats = self._get_xcorrMasks(maskSize) * 2 - 1
        taps = rrcosfilter(0.5, 6, spSym)
        taps = taps / np.sum(taps)
        flen = len(taps)
        templates = [np.convolve(np.repeat(p, spSym), taps)[flen // 2:-flen // 2 + 1] for p in pats]
        return bank_from_templates(templates, Nfft)

    def get_symbolLUT_new(self, maskLen):
        return _nrzs_lut(maskLen)

    def get_symbolLUT2(self, maskLen):
        return None, self.get_symbolLUT_new(maskLen)
