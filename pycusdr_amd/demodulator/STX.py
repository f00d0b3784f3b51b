"""S-band (STX) receive channel: the carrier sits at the configured IF offset, so there is no
Doppler search; strong interference bursts are clipped on the host before the block is transformed.

Entry points and return tuples are those of the reference's ``demodulator.STX.Demodulator``
(reference demodulator/STX.py:6-24).
"""
from .demodulator_base import Demodulator as _HostDriver


class Demodulator(_HostDriver):
    backend = 'STX'

    def uploadAndFindCarrier(self, samples):
        """Clip interference peaks in place (their indices are kept for the trust tagging), then
        forward-FFT the block.  The first, second and fourth results are the constants the
        reference returns in place of a Doppler estimate."""
        self._thresholdInput(samples)
        if self._one_call:
            self.dopplerIdxlast = self.doppOffsetIdx
            self._receive_block(samples, fixed_shift=int(self.doppOffsetIdx))
        else:
            self.uploadToGPU(samples)
        no_estimate = 0
        return no_estimate, no_estimate, self.clippedPeakIPure, no_estimate

    def demodulate(self):
        """Symbols of the last uploaded block at the fixed IF-offset shift."""
        return self.demodulateSTX()
