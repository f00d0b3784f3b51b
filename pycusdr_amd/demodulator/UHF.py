"""UHF receive channel: the block's Doppler shift is unknown, so every block goes through the
matched-filter-bank search before it is demodulated at the shift that was found.

Entry points and return tuples are those of the reference's ``demodulator.UHF.Demodulator``
(reference demodulator/UHF.py:5-20); unlike STX the samples are not peak-clipped first.
"""
from .demodulator_base import Demodulator as _HostDriver


class Demodulator(_HostDriver):
    backend = 'UHF'

    def uploadAndFindCarrier(self, samples, device_ptr=None):
        """Forward FFT of the block on the GPU, then the Doppler search.
        Returns (frequency offset in Hz, search metric, indices of clipped samples, SNR in dB)."""
        if self._one_call:
            return self._estimate_from_block(self._receive_block(samples, device_ptr=device_ptr))
        self.uploadToGPU(samples, device_ptr=device_ptr)
        estimate = self._findUHF(samples)
        return estimate

    def demodulate(self):
        """Symbols of the last uploaded block at the Doppler shift the search settled on.
        Returns (bits, symbol centres mod 256, trust bytes, samples per symbol); arrays are uint8."""
        return self.demodulateUHF()
