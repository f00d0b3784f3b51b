"""GFSK benchmark protocol.  The reference's Bench_GFSK overrides GFSK2.get_filter with the plain
FSK templates (protocol/benchmark/bench_GFSK.py:39-73, "no ISI" comment); reproduced as is."""
from ..FSK2_base import fsk_phase_templates
from ..GFSK2_base import GFSK2
from ..protocolBase import PacketEndDetect, bank_from_templates
from .bench_base import MASKLEN, PACKETLEN, Bench_base


class Bench_GFSK(Bench_base, GFSK2):
    name = 'bench_GFSK'
    packetEndDetectMode = PacketEndDetect.FIXED
    packetLen = PACKETLEN
    numBitsOverlap = MASKLEN * 2
    SUM_ALL_MASKS_PYTHON = True

    def get_filter(self, Nfft, spSym, maskSize):
        return bank_from_templates(fsk_phase_templates(self._get_xcorrMasks(maskSize), spSym, 0.5), Nfft)

    def get_symbolLUT2(self, maskLen):
        return self.centre_bit_lut(maskLen), []
