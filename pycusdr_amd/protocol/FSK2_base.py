"""2-FSK matched-filter generator (reference protocol/FSK2_base.py:17-46)."""
import numpy as np

from .protocolBase import ProtocolBase, bank_from_templates


def fsk_phase_templates(patterns, spSym, cycles_per_symbol):
    """Continuous-phase FSK templates: +/- 2*pi*cycles rad per symbol, starting at -/+ pi/2."""
    ramp = np.linspace(1 / spSym, 1, spSym) * np.pi * 2 * cycles_per_symbol
    out = []
    for bits in patterns:
        pm = bits * 2 - 1
        ph = np.empty(len(pm) * spSym)
        ph[:spSym] = pm[0] * ramp + -1 * pm[0] * np.pi / 2
        for j in range(1, len(pm)):
            ph[j * spSym:(j + 1) * spSym] = ph[j * spSym - 1] + pm[j] * ramp
        out.append(np.exp(1j * ph))
    return out


class FSK2(ProtocolBase):
    name = 'FSK2 Base'

    def get_filter(self, Nfft, spSym, maskSize, nCycles=0.5):
        return bank_from_templates(fsk_phase_templates(self._get_xcorrMasks(maskSize), spSym, nCycles), Nfft)
