"""Gaussian 2-FSK matched-filter generator (reference protocol/GFSK2_base.py:22-60)."""
import numpy as np
from scipy import signal

from ..lib.filters import gaussianFilter
from .protocolBase import ProtocolBase, bank_from_templates

BT = 1.0


def hamming_weighted(templates):
    w = signal.get_window('hamming', len(templates[0]))
    return [t * w for t in templates]


class GFSK2(ProtocolBase):
    name = 'GFSK2 Base'

    def get_filter(self, Nfft, spSym, maskSize):
        taps = gaussianFilter(1, BT, spSym, 4 * spSym) * np.pi / spSym
        flen = len(taps)
        templates = []
        for bits in self._get_xcorrMasks(maskSize):
            phase = np.convolve(np.repeat(bits * 2 - 1, spSym), taps)
            wave = np.exp(1j * np.cumsum(phase))
            templates.append(wave[flen // 2:-flen // 2 + 1])
        return bank_from_templates(hamming_weighted(templates), Nfft)
