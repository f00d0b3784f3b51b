"""UHF back end: forward FFT, Doppler search, demodulation (reference demodulator/UHF.py:5-20)."""
from .demodulator_base import Demodulator as Demodulator_base


class Demodulator(Demodulator_base):

    def uploadAndFindCarrier(self, samples):
        """-> (freqOffset_Hz, metric, clippedPeakIdx, SNR_dB); no input thresholding on UHF."""
        self.uploadToGPU(samples)
        return self._findUHF(samples)

    def demodulate(self):
        """-> (bits uint8[], centres uint8[], trust uint8[], spSym)."""
        return self.demodulateUHF()
