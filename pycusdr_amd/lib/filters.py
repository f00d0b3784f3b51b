"""Pulse-shaping filters used by the filter-bank generators (host, float64, run once at init).

Same call shapes and outputs as the reference's lib/filters.py (rrcosfilter :6-55, gaussianFilter
:58-84); pinned by tests/test_protocols.py against fixtures captured from the reference.
"""
import numpy as np


def rrcosfilter(beta, span, spsym):
    """Root-raised-cosine taps with roll-off ``beta`` over ``span`` symbols at ``spsym`` samples per
    symbol, normalised to unit energy (span*spsym + 1 taps, like MATLAB's rcosdesign 'sqrt')."""
    n = int(span * spsym)
    t = (np.arange(n + 1) - n / 2.0) / spsym
    h = np.empty_like(t)
    eps = np.sqrt(np.finfo(float).eps)
    centre = t == 0
    edge = np.abs(np.abs(4 * beta * t) - 1) < eps
    rest = ~(centre | edge)
    h[centre] = -1 / (np.pi * spsym) * (np.pi * (beta - 1) - 4 * beta)
    if edge.any():
        h[edge] = 1 / (2 * np.pi * spsym) * (
            np.pi * (beta + 1) * np.sin(np.pi * (beta + 1) / (4 * beta))
            - 4 * beta * np.sin(np.pi * (beta - 1) / (4 * beta))
            + np.pi * (beta - 1) * np.cos(np.pi * (beta - 1) / (4 * beta)))
    tr = t[rest]
    h[rest] = -4 * beta / spsym * (np.cos((1 + beta) * np.pi * tr) + np.sin((1 - beta) * np.pi * tr) / (4 * beta * tr)) / (
        np.pi * ((4 * beta * tr) ** 2 - 1))
    return h / np.sqrt(np.sum(h ** 2))


def gaussianFilter(gain, BT, spSym, nTaps):
    """Gaussian pulse (bandwidth-time product ``BT``), ``nTaps`` taps, taps summing to 1/gain."""
    a = np.sqrt(np.log(2) / 2) / BT
    t = np.linspace(-0.5 * nTaps, 0.5 * nTaps - 1, nTaps) / spSym
    ft = np.sqrt(np.pi) / a * np.exp(-(np.pi ** 2 * t ** 2) / a ** 2)
    ft /= np.sum(ft) * gain
    return ft
