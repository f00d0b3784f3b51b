"""GMSK benchmark protocol (reference protocol/benchmark/bench_GMSK.py:17-79)."""
from ...lib.gmskmod import gmskMod
from ..GFSK2_base import hamming_weighted
from ..protocolBase import PacketEndDetect, bank_from_templates
from .bench_base import MASKLEN, PACKETLEN, Bench_base


class Bench_GMSK(Bench_base):
    name = 'bench_GMSK'
    packetEndDetectMode = PacketEndDetect.FIXED
    packetLen = PACKETLEN
    numBitsOverlap = MASKLEN * 2
    SUM_ALL_MASKS_PYTHON = True

    def get_filter(self, Nfft, spSym, maskSize):
        templates = []
        for bits in self._get_xcorrMasks(maskSize):
            wave, _, flen = gmskMod(bits, spSym)
            templates.append(wave[flen // 2:-flen // 2 + 1])
        return bank_from_templates(hamming_weighted(templates), Nfft)

    def get_symbolLUT2(self, maskLen):
        return self.centre_bit_lut(maskLen), []
