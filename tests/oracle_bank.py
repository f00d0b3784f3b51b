"""OracleBank -- a CPU stand-in for pycusdr_amd.mfbank.MFBank built from oracle/ functions.

TEST INFRASTRUCTURE ONLY.  Tests monkeypatch it into the Demodulator to (a) exercise the host logic
end to end on machines without a GPU and (b) produce the oracle's answer for the same block on the
GPU box, so that the HIP path can be compared stage by stage.  Nothing under pycusdr_amd/ imports
this file.
"""
import numpy as np

from oracle import mfbank_oracle as orc


class OracleBank:
    def __init__(self, log2N, num_dopplers, M, window_width=7, sum_all_masks=True,
                 code_search_mask_offset=0, doppler_offset=0, device=0):
        self.N = 1 << int(log2N)
        self.D, self.Doff, self.M = int(num_dopplers), int(doppler_offset), int(M)
        self.Dtot = self.D + self.Doff
        self.W, self.sum_all, self.cs_off = int(window_width), bool(sum_all_masks), int(code_search_mask_offset)
        self.input = np.zeros(self.N, dtype=np.complex64)
        self.masks = self.shifts = self.X = self.xc = self.ds = None

    def close(self):
        pass

    def set_stream(self, s):
        pass

    def set_filters(self, masks):
        masks = np.asarray(masks)
        if masks.dtype != np.complex64:
            raise TypeError('complex64 expected')
        if masks.shape != (self.M, self.N):
            raise ValueError('shape')
        self.masks = masks

    def set_shifts(self, shifts):
        s = np.asarray(shifts, dtype=np.int32)
        if s.size != self.Dtot or s.min() < 0 or s.max() >= self.N:
            raise ValueError('shifts')
        self.shifts = s

    def upload(self, samples=None):
        x = self.input if samples is None else np.asarray(samples, dtype=np.complex64)
        self.X = orc.forward_fft(x)

    def find_carrier(self):
        self.ds = orc.doppler_scores(self.X, self.masks, self.shifts, self.sum_all).astype(np.float32)
        return orc.find_doppler_est(self.ds, self.D, self.Doff, self.sum_all)

    def get_scores(self):
        return self.ds

    # -- the device-pointer calls DopplerShard uses; "device" memory here is host memory of CPU tensors ----------
    @staticmethod
    def _view(ptr, count, dtype=np.float32):
        import ctypes
        nbytes = count * np.dtype(dtype).itemsize
        return np.frombuffer((ctypes.c_char * nbytes).from_address(int(ptr)), dtype=dtype)

    def upload_device(self, ptr):
        self.upload(self._view(ptr, 2 * self.N).view(np.complex64).copy())

    def search_async(self):
        self.ds = orc.doppler_scores(self.X, self.masks, self.shifts, self.sum_all).astype(np.float32)

    def export_scores_async(self, ptr, row_offset):
        self._view(ptr, (row_offset + self.Dtot) * self.M)[row_offset * self.M:] = self.ds.ravel()

    def export_column_async(self, ptr, row_offset):
        self._view(ptr, row_offset + self.Dtot)[row_offset:] = self.ds[:, 0]

    def export_rows_async(self, ptr, dst_row, first_row, nrows, column_only=False):
        if column_only:
            self._view(ptr, dst_row + nrows)[dst_row:] = self.ds[first_row:first_row + nrows, 0]
        else:
            self._view(ptr, (dst_row + nrows) * self.M)[dst_row * self.M:] = self.ds[first_row:first_row + nrows].ravel()

    def pick(self, ptr=None, num=None, offset=None):
        num = self.D if num is None else num
        offset = self.Doff if offset is None else offset
        table = self.ds if not ptr else self._view(ptr, (num + offset) * self.M).reshape(num + offset, self.M)
        return orc.find_doppler_est(table, num, offset, self.sum_all)

    def pick_column(self, ptr, num, offset=0):
        col = self._view(ptr, num + offset).reshape(-1, 1)
        return orc.find_doppler_est(col, num, offset, True)

    def get_spectrum(self, start=0, count=None):
        count = self.N if count is None else count
        idx = (start + np.arange(count)) % self.N
        return self.X[idx]

    def demodulate(self, shift, k_offset, k_len):
        self.xc = orc.demod_xcorr(self.X, self.masks, shift).astype(np.complex64)
        self.env = orc.envelope(self.xc, self.cs_off)
        k, arg, val = orc.code_rate_and_phase(self.env, k_offset, k_len)
        return np.float32(k), np.float32(arg), np.float32(val)

    def find_centres(self, spSym, offset, op, count):
        sym, cen, mag = orc.find_centres(self.xc, spSym, offset, self.W, op)
        return sym[:count], cen[:count], mag[:count]

    def get_xcorr(self):
        return self.xc

    def get_envelope(self):
        return self.env.astype(np.float32)
