"""GMSK modulator used to build the bench_GMSK matched filters (reference lib/gmskmod.py:10-43)."""
import numpy as np

from .filters import gaussianFilter


def gmskMod(bits, spSym, bw=0.5, nTaps=None, gain=1):
    """Returns (waveform, phase, filter_length).  Bits may be 0/1 or +-1."""
    bits = np.asarray(bits)
    if not bits.min() < 0:
        bits = bits * 2 - 1
    if nTaps is None:
        nTaps = 4 * spSym
    taps = gaussianFilter(gain, bw, spSym, nTaps) * np.pi / 2 / spSym
    phase = np.cumsum(np.convolve(taps, np.repeat(bits, spSym)))
    return np.exp(1j * phase), phase, len(taps)
