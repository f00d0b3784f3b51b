"""2-FSK benchmark protocol, +/- pi rad per symbol (reference protocol/benchmark/bench_FSK.py:17-85)."""
from ..FSK2_base import fsk_phase_templates
from ..protocolBase import PacketEndDetect, bank_from_templates
from .bench_base import MASKLEN, PACKETLEN, Bench_base


class Bench_FSK(Bench_base):
    name = 'bench_FSK'
    packetEndDetectMode = PacketEndDetect.FIXED
    packetLen = PACKETLEN
    numBitsOverlap = MASKLEN * 2
    SUM_ALL_MASKS_PYTHON = True

    def get_filter(self, Nfft, spSym, maskSize):
        return bank_from_templates(fsk_phase_templates(self._get_xcorrMasks(maskSize), spSym, 0.5), Nfft)

    def get_symbolLUT2(self, maskLen):
        return self.centre_bit_lut(maskLen), []
