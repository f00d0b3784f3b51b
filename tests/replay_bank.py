"""ReplayBank -- a stand-in for pycusdr_amd.mfbank.MFBank that computes nothing: every call that would return a device
result hands out the value the test queued for it, and records what it was called with.

TEST INFRASTRUCTURE ONLY.  It puts this repo's host driver (``Demodulator``) in the position the recording fake of
tests/golden/make_golden_host.py put the reference's: the same injected device results go in at the same points, and the
outputs of the two host codes are compared bit for bit (tests/test_pins_host.py).  Nothing under pycusdr_amd/ imports it.
"""
import numpy as np


class ReplayBank:
    def __init__(self, log2N, num_dopplers, M, window_width=7, sum_all_masks=True, code_search_mask_offset=0,
                 doppler_offset=0, device=0):
        self.N = 1 << int(log2N)
        self.D, self.Doff, self.M = int(num_dopplers), int(doppler_offset), int(M)
        self.Dtot = self.D + self.Doff
        self.input = np.zeros(self.N, dtype=np.complex64)
        self.pick = self.triple = self.sym = self.cen = self.mag = self.X = None
        self.calls = []

    def close(self):
        pass

    def set_filters(self, masks):
        self.filters_shape = np.asarray(masks).shape

    def set_shifts(self, shifts):
        self.shifts = np.asarray(shifts, dtype=np.int32).copy()

    def upload(self, samples=None):
        if getattr(self, 'on_upload', None) is not None:      # a test's hook: sees the block, queues the next device results
            self.on_upload(self.input if samples is None else samples)
        self.calls.append(('upload',))

    def find_carrier(self):
        self.calls.append(('find_carrier',))
        return np.float32(self.pick[0]), np.float32(self.pick[1])

    def get_spectrum(self, start=0, count=None):
        count = self.N if count is None else count
        return self.X[(start + np.arange(count)) % self.N]

    def demodulate(self, shift, k_offset, k_len):
        self.calls.append(('demodulate', int(shift), int(k_offset), int(k_len)))
        return np.float32(self.triple[0]), np.float32(self.triple[1]), np.float32(self.triple[2])

    def find_centres(self, spSym, offset, op, count):
        self.calls.append(('find_centres', np.float32(spSym), np.float32(offset), int(op), int(count)))
        return self.sym[:count].copy(), self.cen[:count].copy(), self.mag[:count].copy()
