"""S-band (STX) back end: peak clipping + forward FFT, fixed shift, no Doppler search
(reference demodulator/STX.py:6-24)."""
from .demodulator_base import Demodulator as Demodulator_base


class Demodulator(Demodulator_base):

    def uploadAndFindCarrier(self, samples):
        self._thresholdInput(samples)
        self.uploadToGPU(samples)
        return 0, 0, self.clippedPeakIPure, 0

    def demodulate(self):
        return self.demodulateSTX()
