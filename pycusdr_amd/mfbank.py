"""MFBank -- Python handle over the libmfbank C ABI (one per process, like the reference's CUDA
context per Demodulator_process, reference demodulator_base.py:177-181).

Only numpy arrays and plain ints/floats cross this layer; all arithmetic of the hot path happens
in the HIP kernels behind it.
"""
import ctypes as C

import numpy as np

from . import _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# struct BlockScalars of csrc/small_kernels.hpp (the head of every block record), field for field
BLOCK_SCALARS = np.dtype({
    'names': ['frac', 'spSym', 'codeOffset', 'pick', 'cr', 'spSymF', 'offsetF', 'shift', 'low', 'high', 'pick_valid', 'count',
              'rate_fallback', 'band', 'band_len', 'a13_status', 'a13_start', 'a13_end', 'a13_nwin', 'a13_noerr', 'a13_npost',
              'a13_nend', 'sync_valid', 'sync_count', 'a13_prev_npost'],
    'formats': ['<f8', '<f8', '<f8', ('<f4', 2), ('<f4', 3), '<f4', '<f4', '<i4', '<i4', '<i4', '<i4', '<i4', '<i4', ('<i4', (2, 2, 2)),
                ('<i4', 2), '<i4', '<i4', '<i4', '<i4', '<i4', '<i4', '<i4', '<i4', ('<i4', 2), '<i4'],
    'offsets': [0, 8, 16, 24, 32, 44, 48, 52, 56, 60, 64, 68, 72, 76, 108, 116, 120, 124, 128, 132, 136, 140, 144, 148, 156],
    'itemsize': 168})


class HostCopy:
    """The library's host copy worker (mfb_hostcopy_*): ``submit(dst, dst_off, src)`` queues ``dst[dst_off:dst_off + len(src)] = src``
    for a thread of its own and returns at once; ``drain()`` returns when every queued copy has been made.  Arrays handed to
    ``submit`` are kept alive until then.  Plain host memory: no GPU involved."""

    def __init__(self):
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.mfb_hostcopy_create(C.byref(h)), 'mfb_hostcopy_create')
        self._h = h
        self._held = []
        self._base = {}              # id(dst) -> (address, itemsize): asking numpy for it costs as much as a small copy

    def submit(self, dst, dst_off, src):
        n = len(src)
        if not n:
            return
        key = id(dst)
        base = self._base.get(key)
        if base is None or base[2] is not dst:
            if not dst.flags.c_contiguous or not dst.flags.writeable:
                raise ValueError('destination must be a writable contiguous array')
            base = self._base[key] = (dst.ctypes.data, dst.itemsize, dst)
        if src.dtype != dst.dtype or not src.flags.c_contiguous:
            src = np.ascontiguousarray(src, dtype=dst.dtype)
        if dst_off < 0 or dst_off + n > len(dst):
            raise IndexError('copy outside the destination')
        self._held.append(src)
        rc = self._lib.mfb_hostcopy_submit(self._h, base[0] + dst_off * base[1], src.__array_interface__['data'][0], n * base[1])
        _lib.check(rc, 'mfb_hostcopy_submit')

    def drain(self):
        if self._held:
            _lib.check(self._lib.mfb_hostcopy_drain(self._h), 'mfb_hostcopy_drain')
            del self._held[:]

    def close(self):
        if self._h is not None:
            self._lib.mfb_hostcopy_destroy(self._h)
            self._h = None
            del self._held[:]
            self._base.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:        # noqa: BLE001 -- interpreter shutdown
            pass


class BatchRecord:
    """A finished batch of blocks as it came off the device (mfb_receive_blocks_end_record): ``nb`` records in one buffer, read
    in place.  Scalars come as one Python list per field (``s['count'][b]``); arrays as 2-D views, block b in row b."""

    def __init__(self, buf, lay, searched):
        nb, rec, n = lay.nblocks, lay.record_bytes, lay.symbols
        self.nb, self.searched, self.bcap, self.mode = nb, searched, lay.band_capacity, lay.mode
        if lay.scalars_bytes != BLOCK_SCALARS.itemsize:
            raise RuntimeError('the library\'s BlockScalars layout differs from this binding\'s')
        a = buf[:nb * rec].reshape(nb, rec)
        heads = a[:, :BLOCK_SCALARS.itemsize].copy().view(BLOCK_SCALARS).reshape(nb)
        self.s = {k: heads[k].tolist() for k in BLOCK_SCALARS.names}
        if not searched:
            self.s['shift'] = [int(lay.fixed_shift)] * nb

        def arr(off, count, dt):
            return a[:, off:off + count * np.dtype(dt).itemsize].view(dt)
        self.sym, self.cen, self.mag = arr(lay.off_sym, n, np.int32), arr(lay.off_cen, n, np.int32), arr(lay.off_mag, n, np.float32)
        self.bands = arr(lay.off_bands, 2 * lay.band_capacity, np.complex64).reshape(nb, 2, lay.band_capacity) if lay.band_capacity else None
        self.stages = bool(lay.stream_stages)
        if self.stages:
            self.bits, self.cen8, self.trust = (arr(lay.off_bits, n, np.uint8), arr(lay.off_centres_u8, n, np.uint8),
                                                arr(lay.off_trust, n, np.uint8))
            self.post, self.end = a[:, lay.off_post:lay.off_post + 512], a[:, lay.off_end:lay.off_end + 32]
            self.templates, mh = lay.templates, lay.max_hits
            self.hits = arr(lay.off_hits, 2 * 2 * mh, np.int32).reshape(nb, 2, 2, mh)      # [block][template][idx | score][hit]
            self.max_hits = mh
            # [block][candidate]{a_rel, valid, n[2], idx[2][eh], score[2][eh]}: the leading positions of would-be stash streams
            ec, eh = lay.edge_candidates, lay.edge_hits
            self.edges = arr(lay.off_edges, ec * (4 + 4 * eh), np.int32).reshape(nb, ec, 4 + 4 * eh) if ec else None
            self.edge_hits = eh

    def block(self, b):
        """Block b as the dict ``receive_block`` returns."""
        s, n = self.s, self.s['count'][b]
        d = {'pick': (np.float32(s['pick'][b][0]), np.float32(s['pick'][b][1])), 'pick_valid': bool(s['pick_valid'][b]),
             'shift': int(s['shift'][b]), 'low': s['low'][b], 'high': s['high'][b], 'frac': s['frac'][b],
             'cr': tuple(np.float32(v) for v in s['cr'][b]), 'spSym': s['spSym'][b], 'codeOffset': s['codeOffset'][b],
             'rate_fallback': bool(s['rate_fallback'][b]), 'symbols': self.sym[b, :n], 'centres': self.cen[b, :n],
             'magnitudes': self.mag[b, :n], 'bands': None}
        l0, l1 = s['band_len'][b]
        if self.searched and self.bands is not None and l0 <= self.bcap and l1 <= self.bcap:
            d['bands'] = (self.bands[b, 0, :l0], self.bands[b, 1, :l1])
        return d


class MFBank:
    def __init__(self, log2N, num_dopplers, M, window_width=7, sum_all_masks=True,
                 code_search_mask_offset=0, doppler_offset=0, device=0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.N = 1 << int(log2N)
        self.D = int(num_dopplers)
        self.Doff = int(doppler_offset)
        self.Dtot = self.D + self.Doff
        self.M = int(M)
        self.device = int(device)
        _lib.check(self._lib.mfb_create(C.byref(self._h), self.device, int(log2N), self.D, self.Doff, self.M,
                                        int(window_width), int(bool(sum_all_masks)), int(code_search_mask_offset)),
                   'mfb_create')
        buf = C.POINTER(C.c_float)()
        _lib.check(self._lib.mfb_input_buffer(self._h, C.byref(buf)), 'mfb_input_buffer')
        # writable complex64 view of the page-locked input buffer owned by the library
        self.input = np.ctypeslib.as_array(buf, shape=(2 * self.N,)).view(np.complex64)

    # -- lifetime --------------------------------------------------------------------------------
    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self.input = None
            self._input2 = None
            self._wins, self._win_key = None, None
            self._lib.mfb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration ---------------------------------------------------------------------------
    def set_stream(self, hip_stream):
        _lib.check(self._lib.mfb_set_stream(self._h, C.c_void_p(hip_stream or 0)), 'mfb_set_stream')

    def set_tuning(self, doppler_chunk=0, masks_per_block=0, rows_per_block=0, jsplit=0):
        _lib.check(self._lib.mfb_set_tuning(self._h, int(doppler_chunk), int(masks_per_block), int(rows_per_block),
                                            int(jsplit)), 'mfb_set_tuning')

    def get_tuning(self):
        a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _lib.check(self._lib.mfb_get_tuning(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)), 'mfb_get_tuning')
        return a.value, b.value, c.value, d.value

    def get_info(self):
        """(N1, N2, unique filter rows transformed by the search)."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        _lib.check(self._lib.mfb_get_info(self._h, C.byref(a), C.byref(b), C.byref(c)), 'mfb_get_info')
        return a.value, b.value, c.value

    PATHS = {'auto': 0, 'twopass': 1, 'segment': 2}

    def set_search_path(self, path='auto', log2L=0, wg_per_cu=0, filters_per_pass=0):
        """Choose between the single-pass overlap-save search ('segment': short filters, no HBM
        intermediate) and the length-N two-pass transforms ('twopass'); 'auto' takes the segment path
        whenever the filter bank has a short impulse response.  Raises ValueError if 'segment' is
        demanded for a bank that does not allow it."""
        _lib.check(self._lib.mfb_set_search_path(self._h, self.PATHS[path] if isinstance(path, str) else int(path),
                                                 int(log2L), int(wg_per_cu), int(filters_per_pass)), 'mfb_set_search_path')

    def set_search_basis(self, basis='filters'):
        """'filters' (default): the search transforms every unique filter, as the reference does.  'span'
        (opt-in; SUM_ALL_MASKS on the segment path): it transforms an orthogonalised basis of the bank's span --
        the same doppSum to fp32 rounding from rank(bank) instead of M inverse transforms per segment."""
        _lib.check(self._lib.mfb_set_search_basis(self._h, {'filters': 0, 'span': 1}[basis]), 'mfb_set_search_basis')

    def get_search_basis(self):
        """(basis in force, filters transformed per Doppler bin)."""
        b, n = C.c_int(), C.c_int()
        _lib.check(self._lib.mfb_get_search_basis(self._h, C.byref(b), C.byref(n)), 'mfb_get_search_basis')
        return ('span' if b.value == 1 else 'filters'), n.value

    def set_search_mode(self, mode='transforms'):
        """'transforms' (default): the search runs the matched-filter bank.  'energy' (opt-in): Parseval's identity --
        doppSum from the block's power spectrum and the filters' energy spectrum, D.N multiply-adds and no inverse
        transform; the same table to fp32 rounding.  The demodulation stage is unaffected."""
        _lib.check(self._lib.mfb_set_search_mode(self._h, {'transforms': 0, 'energy': 1}[mode]), 'mfb_set_search_mode')

    def get_search_mode(self):
        m = C.c_int()
        _lib.check(self._lib.mfb_get_search_mode(self._h, C.byref(m)), 'mfb_get_search_mode')
        return 'energy' if m.value == 1 else 'transforms'

    def get_search_path(self):
        """dict(path, log2L, taps, valid_per_segment, segments) in force."""
        v = [C.c_int() for _ in range(5)]
        _lib.check(self._lib.mfb_get_search_path(self._h, *[C.byref(x) for x in v]), 'mfb_get_search_path')
        names = {1: 'twopass', 2: 'segment'}
        return dict(path=names.get(v[0].value, 'twopass'), log2L=v[1].value, taps=v[2].value,
                    valid_per_segment=v[3].value, segments=v[4].value)

    def set_cu_share(self, part, parts):
        """This handle's launches on part ``part`` of ``parts`` equal parts of the device's compute units (mfb_set_cu_share):
        for several demodulator instances on one device.  ``parts = 1``: the whole device."""
        _lib.check(self._lib.mfb_set_cu_share(self._h, int(part), int(parts)), 'mfb_set_cu_share')

    def get_search_info(self):
        """{'filter_side': bool, 'bins_per_forward': int} -- whether the segment search transforms a segment once for several
        bins, the Doppler shift sitting on the filters' side (mfb_get_search_info)."""
        f, b = C.c_int(), C.c_int()
        _lib.check(self._lib.mfb_get_search_info(self._h, C.byref(f), C.byref(b)), 'mfb_get_search_info')
        return {'filter_side': bool(f.value), 'bins_per_forward': int(b.value)}

    def xcorr(self, a, b):
        """ifft(fft(a, N) * conj(fft(b, N))) for real sequences a, b (reference lib/customXCorr.py:5-18);
        N is this handle's block length.  The handle's filters and input are invalidated."""
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        out = np.empty(self.N, dtype=np.complex64)
        _lib.check(self._lib.mfb_xcorr(self._h, _ptr(a), a.size, _ptr(b), b.size, _ptr(out)), 'mfb_xcorr')
        return out

    def set_filters(self, masks):
        masks = np.asarray(masks)
        if masks.ndim != 2:
            raise ValueError(f'filter bank must be 2-D (M, N), got shape {masks.shape}')
        if masks.dtype != np.complex64:
            raise TypeError(f'Datatype of masks {masks.dtype}, expected complex64')
        masks = np.ascontiguousarray(masks)
        _lib.check(self._lib.mfb_set_filters(self._h, _ptr(masks), masks.shape[0], masks.shape[1]), 'mfb_set_filters')

    def set_shifts(self, shifts):
        s = np.ascontiguousarray(np.asarray(shifts), dtype=np.int32)
        _lib.check(self._lib.mfb_set_shifts(self._h, _ptr(s), s.size), 'mfb_set_shifts')

    # -- data in ---------------------------------------------------------------------------------
    def upload(self, samples=None):
        """Forward-FFT the pinned input buffer (samples None or the buffer itself) or a host array."""
        if samples is None or samples is self.input:
            _lib.check(self._lib.mfb_upload(self._h), 'mfb_upload')
            return
        s = np.ascontiguousarray(samples, dtype=np.complex64)
        _lib.check(self._lib.mfb_upload_from(self._h, _ptr(s), s.size), 'mfb_upload_from')

    def upload_device(self, dev_ptr):
        _lib.check(self._lib.mfb_upload_device(self._h, C.c_void_p(int(dev_ptr))), 'mfb_upload_device')

    # -- Doppler search --------------------------------------------------------------------------
    def search_async(self):
        _lib.check(self._lib.mfb_search_async(self._h), 'mfb_search_async')

    def export_scores_async(self, dev_ptr, row_offset):
        _lib.check(self._lib.mfb_export_scores_async(self._h, C.c_void_p(int(dev_ptr)), int(row_offset)),
                   'mfb_export_scores_async')

    def pick(self, dev_scores=None, num=None, offset=None):
        res = (C.c_float * 2)()
        num = self.D if num is None else int(num)
        offset = self.Doff if offset is None else int(offset)
        _lib.check(self._lib.mfb_pick(self._h, C.c_void_p(int(dev_scores) if dev_scores else 0), num, offset, res), 'mfb_pick')
        return np.float32(res[0]), np.float32(res[1])

    def export_column_async(self, dev_ptr, row_offset):
        _lib.check(self._lib.mfb_export_column_async(self._h, C.c_void_p(int(dev_ptr)), int(row_offset)),
                   'mfb_export_column_async')

    def export_rows_async(self, dev_ptr, dst_row, first_row, nrows, column_only=False):
        """Rows [first_row, first_row + nrows) of doppSum (whole rows, or column 0 alone) to rows dst_row... of a
        device array."""
        _lib.check(self._lib.mfb_export_rows_async(self._h, C.c_void_p(int(dev_ptr)), int(dst_row), int(first_row), int(nrows),
                                                   int(bool(column_only))), 'mfb_export_rows_async')

    def pick_column(self, dev_column, num, offset=0):
        res = (C.c_float * 2)()
        _lib.check(self._lib.mfb_pick_column(self._h, C.c_void_p(int(dev_column)), int(num), int(offset), res), 'mfb_pick_column')
        return np.float32(res[0]), np.float32(res[1])

    def find_carrier(self):
        res = (C.c_float * 2)()
        _lib.check(self._lib.mfb_find_carrier(self._h, res), 'mfb_find_carrier')
        return np.float32(res[0]), np.float32(res[1])

    def get_scores(self):
        out = np.empty((self.Dtot, self.M), dtype=np.float32)
        _lib.check(self._lib.mfb_get_scores(self._h, _ptr(out)), 'mfb_get_scores')
        return out

    def get_spectrum(self, start=0, count=None):
        count = self.N if count is None else int(count)
        out = np.empty(count, dtype=np.complex64)
        _lib.check(self._lib.mfb_get_spectrum(self._h, _ptr(out), int(start), count), 'mfb_get_spectrum')
        return out

    # -- demodulation ----------------------------------------------------------------------------
    def demodulate(self, shift, k_offset, k_len):
        res = (C.c_float * 3)()
        _lib.check(self._lib.mfb_demodulate(self._h, int(shift), int(k_offset), int(k_len), res), 'mfb_demodulate')
        return np.float32(res[0]), np.float32(res[1]), np.float32(res[2])

    def find_centres(self, spSym, offset, op, count):
        count = int(count)
        sym = np.empty(count, dtype=np.int32)
        cen = np.empty(count, dtype=np.int32)
        mag = np.empty(count, dtype=np.float32)
        _lib.check(self._lib.mfb_find_centres(self._h, C.c_float(spSym), C.c_float(offset), int(op), count,
                                              _ptr(sym), _ptr(cen), _ptr(mag)), 'mfb_find_centres')
        return sym, cen, mag

    BAND_CAPACITY = 1024       # complex64 elements per SNR window delivered with the block (longer ones: get_spectrum);
                               # an instance may set its own before the first block (Demodulator: from the bin spacing)

    SOURCES = {'pinned': 0, 'device': 1, 'uploaded': 2, 'pinned2': 3, 'window': 4, 'window2': 5}

    def _block_params(self, k_offset, k_len, spsym_min, op, snr_window, fixed_shift, source, device_ptr, block_stride=0):
        P = _lib.BlockParams()
        P.block_stride = int(block_stride)
        P.mode = 0 if fixed_shift is None else 1
        P.input = self.SOURCES[source]
        P.device_block = C.c_void_p(int(device_ptr)) if device_ptr else None
        P.fixed_shift = 0 if fixed_shift is None else int(fixed_shift)
        P.k_offset, P.k_len, P.spsym_min, P.op, P.snr_window = int(k_offset), int(k_len), int(spsym_min), int(op), int(snr_window)
        P.max_symbols, P.band_capacity = self.N // 2, (self.BAND_CAPACITY if fixed_shift is None else 0)
        return P

    def _block_arrays(self):
        if getattr(self, '_blk', None) is None:
            cap = self.N // 2
            self._blk = (np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.float32),
                         np.empty((2, self.BAND_CAPACITY), np.complex64))
        return self._blk

    def _block_result(self, R, searched):
        sym, cen, mag, bands = self._block_arrays()
        n = R.count
        out = {'pick': (np.float32(R.pick[0]), np.float32(R.pick[1])), 'pick_valid': bool(R.pick_valid), 'shift': int(R.shift),
               'low': int(R.low), 'high': int(R.high), 'frac': float(R.frac),
               'cr': (np.float32(R.cr[0]), np.float32(R.cr[1]), np.float32(R.cr[2])), 'spSym': float(R.spSym),
               'codeOffset': float(R.codeOffset), 'rate_fallback': bool(R.rate_fallback),
               'symbols': sym[:n].copy(), 'centres': cen[:n].copy(), 'magnitudes': mag[:n].copy(), 'bands': None}
        if searched and R.band_len[0] <= self.BAND_CAPACITY and R.band_len[1] <= self.BAND_CAPACITY:
            out['bands'] = (bands[0, :R.band_len[0]].copy(), bands[1, :R.band_len[1]].copy())
        return out

    def receive_block(self, k_offset, k_len, spsym_min, op=0, snr_window=5, fixed_shift=None, source='pinned', device_ptr=None):
        """The whole device side of one block in ONE library call with one synchronisation (mfb_receive_block): upload
        (``source``: 'pinned' = the library's input buffer, 'device' = ``device_ptr``, 'uploaded' = an earlier upload),
        Doppler search + pick + shift interpolation (or ``fixed_shift`` for the STX back end), the spectrum windows of
        computeSNR, matched filters, rate/phase, symbol centres.  Returns a dict of plain numbers and numpy arrays."""
        P, R = self._block_params(k_offset, k_len, spsym_min, op, snr_window, fixed_shift, source, device_ptr), _lib.BlockResult()
        sym, cen, mag, bands = self._block_arrays()
        _lib.check(self._lib.mfb_receive_block(self._h, C.byref(P), C.byref(R), _ptr(sym), _ptr(cen), _ptr(mag), _ptr(bands)),
                   'mfb_receive_block')
        return self._block_result(R, fixed_shift is None)

    def debug_block_scalars(self, picks, triples, spsym_min, snr_window=5, max_symbols=None):
        """Test seam (mfb_debug_block_scalars): the one-call path's two float64 device stages on injected picks
        float32[n][2] and rate triples float32[n][3].  Returns a list of dicts with the scalar fields of ``receive_block``
        plus 'spSymF' / 'offsetF' (what findCentres is launched with) and 'band_pieces' int32[2][2][2]."""
        picks = np.ascontiguousarray(picks, dtype=np.float32).reshape(-1, 2)
        triples = np.ascontiguousarray(triples, dtype=np.float32).reshape(-1, 3)
        n = len(picks)
        if len(triples) != n:
            raise ValueError('one rate triple per pick')
        R = (_lib.BlockResult * n)()
        args = np.empty((n, 2), np.float32)
        pieces = np.empty((n, 2, 2, 2), np.int32)
        _lib.check(self._lib.mfb_debug_block_scalars(self._h, n, _ptr(picks), _ptr(triples), int(spsym_min), int(snr_window),
                                                     int(self.N // 2 if max_symbols is None else max_symbols), R, _ptr(args),
                                                     _ptr(pieces)), 'mfb_debug_block_scalars')
        return [{'pick': (np.float32(r.pick[0]), np.float32(r.pick[1])), 'pick_valid': bool(r.pick_valid), 'shift': int(r.shift),
                 'low': int(r.low), 'high': int(r.high), 'frac': float(r.frac),
                 'cr': (np.float32(r.cr[0]), np.float32(r.cr[1]), np.float32(r.cr[2])), 'spSym': float(r.spSym),
                 'codeOffset': float(r.codeOffset), 'count': int(r.count), 'rate_fallback': bool(r.rate_fallback),
                 'band_len': (int(r.band_len[0]), int(r.band_len[1])), 'spSymF': args[i, 0], 'offsetF': args[i, 1],
                 'band_pieces': pieces[i]} for i, r in enumerate(R)]

    @property
    def flights(self):
        """Blocks begun and not yet collected."""
        return len(getattr(self, '_flying', ()))

    @property
    def input2(self):
        """The second page-locked input buffer: while the device works on the block in one buffer, the caller assembles
        the next block in the other (``begin_block(..., source='pinned2')``)."""
        if getattr(self, '_input2', None) is None:
            buf = C.POINTER(C.c_float)()
            _lib.check(self._lib.mfb_input_buffer2(self._h, C.byref(buf)), 'mfb_input_buffer2')
            self._input2 = np.ctypeslib.as_array(buf, shape=(2 * self.N,)).view(np.complex64)
        return self._input2

    def begin_block(self, slot, k_offset, k_len, spsym_min, op=0, snr_window=5, fixed_shift=None, source='pinned', device_ptr=None):
        """First half of ``receive_block``: enqueue everything (read-back included) and return at once.  Up to two blocks
        (``slot`` 0 / 1) may be in flight."""
        P = self._block_params(k_offset, k_len, spsym_min, op, snr_window, fixed_shift, source, device_ptr)
        _lib.check(self._lib.mfb_receive_block_begin(self._h, C.byref(P), int(slot)), 'mfb_receive_block_begin')
        self._searched = getattr(self, '_searched', {})
        self._searched[int(slot)] = fixed_shift is None
        self._flying = getattr(self, '_flying', set()) | {int(slot)}

    def end_block(self, slot):
        """Second half: wait for the block begun in ``slot`` and return its results (same dict as ``receive_block``)."""
        R = _lib.BlockResult()
        sym, cen, mag, bands = self._block_arrays()
        try:
            _lib.check(self._lib.mfb_receive_block_end(self._h, int(slot), C.byref(R), _ptr(sym), _ptr(cen), _ptr(mag), _ptr(bands)),
                       'mfb_receive_block_end')
        finally:            # collected, or failed: either way the slot holds no block any more
            self._flying = getattr(self, '_flying', set()) - {int(slot)}
        return self._block_result(R, self._searched.get(int(slot), True))

    # -- B consecutive blocks per call -----------------------------------------------------------
    def windows(self, max_blocks, block_stride):
        """The two page-locked sample windows of the batched block path (mfb_window_buffer): each holds ``max_blocks``
        consecutive blocks of the stream as ``max_blocks * block_stride + (N - block_stride)`` complex64 samples -- block b
        starts at ``b * block_stride``, neighbours share their overlap.  Returns (window 0, window 1) as writable numpy
        views; asking for another geometry re-allocates them."""
        key = (int(max_blocks), int(block_stride))
        if getattr(self, '_win_key', None) != key:
            n = key[0] * key[1] + (self.N - key[1])
            wins = []
            for which in (0, 1):
                buf = C.POINTER(C.c_float)()
                _lib.check(self._lib.mfb_window_buffer(self._h, which, key[0], key[1], C.byref(buf)), 'mfb_window_buffer')
                wins.append(np.ctypeslib.as_array(buf, shape=(2 * n,)).view(np.complex64))
            self._wins, self._win_key = tuple(wins), key
        return self._wins

    def begin_blocks(self, slot, nblocks, k_offset, k_len, spsym_min, op=0, snr_window=5, fixed_shift=None, source='window',
                     device_ptr=None, block_stride=0):
        """``begin_block`` for ``nblocks`` consecutive blocks of a window (mfb_receive_blocks_begin): one set of launches, one
        read-back; returns at once.  ``source``: 'window' / 'window2' (``windows()``) or 'device' (``device_ptr`` = the window in
        device memory, ``block_stride`` required)."""
        P = self._block_params(k_offset, k_len, spsym_min, op, snr_window, fixed_shift, source, device_ptr, block_stride)
        _lib.check(self._lib.mfb_receive_blocks_begin(self._h, C.byref(P), int(nblocks), int(slot)), 'mfb_receive_blocks_begin')
        self._searched = getattr(self, '_searched', {})
        self._searched[int(slot)] = fixed_shift is None
        self._batch = getattr(self, '_batch', {})
        self._batch[int(slot)] = (int(nblocks), int(k_offset) + int(k_len) + 1)
        self._flying = getattr(self, '_flying', set()) | {int(slot)}

    def get_batch_scores(self, block):
        """``get_scores`` for block ``block`` of the batch begun last (collected, and no later batch begun)."""
        out = np.empty((self.Dtot, self.M), dtype=np.float32)
        _lib.check(self._lib.mfb_get_batch_scores(self._h, int(block), _ptr(out)), 'mfb_get_batch_scores')
        return out

    def set_batch_overlap(self, on=True):
        """A batch as two parts on two streams (mfb_set_batch_overlap): the next batch's search runs beside this batch's matched
        filters, rate / phase, centres and integer stages.  For callers that keep two batches in flight and are not bound by
        their own per-block work; off by default."""
        _lib.check(self._lib.mfb_set_batch_overlap(self._h, 1 if on else 0), 'mfb_set_batch_overlap')

    def end_blocks_record(self, slot):
        """Wait for the batch begun in ``slot`` and return it as a ``BatchRecord`` (one copy, read in place)."""
        slot = int(slot)
        lay = _lib.RecordLayout()
        buf = np.empty(getattr(self, '_recbuf_need', 1 << 16), np.uint8)      # a fresh buffer per batch: the views handed out keep it alive
        rc = self._lib.mfb_receive_blocks_end_record(self._h, slot, _ptr(buf), buf.size, C.byref(lay))
        if rc == _lib.MFB_ERR_ARG and lay.nblocks > 0 and lay.record_bytes > 0:
            # too small for this batch: the library said what it needs and left the batch in flight
            self._recbuf_need = int(lay.nblocks) * int(lay.record_bytes)
            buf = np.empty(self._recbuf_need, np.uint8)
            rc = self._lib.mfb_receive_blocks_end_record(self._h, slot, _ptr(buf), buf.size, C.byref(lay))
        if rc == _lib.MFB_OK or rc == _lib.MFB_ERR_STATE:
            # taken (or never there): the slot is free.  Any other status leaves the batch in flight, and `flights` must go on
            # saying so (computeSNR's stale-spectrum guard reads it)
            self._flying = getattr(self, '_flying', set()) - {slot}
        _lib.check(rc, 'mfb_receive_blocks_end_record')
        return BatchRecord(buf, lay, self._searched.get(slot, True))

    def end_blocks(self, slot):
        """Wait for the batch begun in ``slot``; one dict per block, each exactly what ``receive_block`` returns for that block
        alone (the arrays are views into the batch's record -- no further copies)."""
        rec = self.end_blocks_record(slot)
        return [rec.block(b) for b in range(rec.nb)]

    # -- the integer stages behind the symbol decisions, on the device (batches) ---------------------
    def set_stream_stages(self, overlap_samples, overlap_offset, match_threshold, error_threshold, bit_lut=None, nrzs_lut=None,
                          templates=(), thresholds=(), bits_overlap=0):
        """A12 (bit LUT uint8[rows] of 0 / 1, or the 3-D NRZ-S LUT int[rows][2][successors]), A13 (block-overlap alignment) and --
        with ``templates`` -- A14 (the decoder's searches on the stream without a stash) for the blocks of a batch
        (mfb_set_stream_stages).  ``None`` for both LUTs switches the stages off."""
        if bit_lut is None and nrzs_lut is None:
            _lib.check(self._lib.mfb_set_stream_stages(self._h, None), 'mfb_set_stream_stages')
            self._stages = False
            return
        P = _lib.StreamParams()
        P.overlap_samples, P.overlap_offset, P.match_threshold, P.error_threshold = (int(overlap_samples), int(overlap_offset),
                                                                                     int(match_threshold), int(error_threshold))
        if bit_lut is not None:
            lut = np.ascontiguousarray(bit_lut, dtype=np.uint8)
            P.lut_mode, P.lut_rows, P.lut_successors = 1, lut.size, 0
        else:
            lut = np.ascontiguousarray(nrzs_lut, dtype=np.int32)
            if lut.ndim != 3 or lut.shape[1] != 2:
                raise ValueError('NRZ-S LUT: int[rows][2][successors]')
            P.lut_mode, P.lut_rows, P.lut_successors = 2, lut.shape[0], lut.shape[2]
        P.lut = lut.ctypes.data
        tis = [np.ascontiguousarray(t, dtype=np.int8) for t in templates]
        for t, ti in zip(templates, tis):
            if not np.array_equal(ti, np.asarray(t)) or np.abs(ti).max(initial=0) > 1:
                raise ValueError('templates: taps in {-1, 0, +1}')
        P.num_templates, P.bits_overlap = len(tis), int(bits_overlap)
        packed = np.concatenate(tis) if tis else np.zeros(0, np.int8)
        for k, ti in enumerate(tis):
            P.template_taps[k] = ti.size
            P.template_thresholds[k] = int(np.ceil(thresholds[k]))
        P.templates = packed.ctypes.data if tis else None
        _lib.check(self._lib.mfb_set_stream_stages(self._h, C.byref(P)), 'mfb_set_stream_stages')
        self._stages = True

    def debug_stream_stages(self, counts, sym, cen, mag):
        """Test seam (mfb_debug_stream_stages): the stream-stage kernels on injected symbol decisions int32 [nb][symbols] (centres,
        float32 magnitudes alike; ``counts[b]`` valid entries) in front of the seeded state.  Returns a ``BatchRecord``."""
        sym = np.ascontiguousarray(sym, dtype=np.int32)
        cen = np.ascontiguousarray(cen, dtype=np.int32)
        mag = np.ascontiguousarray(mag, dtype=np.float32)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        nb, n = sym.shape
        lay = _lib.RecordLayout()
        buf = np.empty(nb * (8192 + 16 * n), np.uint8)
        _lib.check(self._lib.mfb_debug_stream_stages(self._h, nb, n, _ptr(counts), _ptr(sym), _ptr(cen), _ptr(mag), _ptr(buf), buf.size,
                                                     C.byref(lay)), 'mfb_debug_stream_stages')
        return BatchRecord(buf, lay, True)

    def stream_seed(self, post, end, ring=None):
        """The state the next batch starts from: the previous block's tail (``post``: the bits behind its window, ``end``: the last
        overlap_offset + 1 bits inside it) and the last bits_overlap bits of the decoder's stream (``ring``; None: unknown)."""
        post = np.ascontiguousarray(post, dtype=np.uint8)
        end = np.ascontiguousarray(end, dtype=np.uint8)
        ring = np.zeros(0, np.uint8) if ring is None else np.ascontiguousarray(ring, dtype=np.uint8)
        _lib.check(self._lib.mfb_stream_seed(self._h, _ptr(post), post.size, _ptr(end), end.size, _ptr(ring), ring.size), 'mfb_stream_seed')

    def get_xcorr(self):
        out = np.empty((self.M, self.N), dtype=np.complex64)
        _lib.check(self._lib.mfb_get_xcorr(self._h, _ptr(out)), 'mfb_get_xcorr')
        return out

    def get_envelope(self):
        out = np.empty(self.N, dtype=np.float32)
        _lib.check(self._lib.mfb_get_envelope(self._h, _ptr(out)), 'mfb_get_envelope')
        return out

    # -- timing ----------------------------------------------------------------------------------
    def timer_start(self):
        _lib.check(self._lib.mfb_timer_start(self._h), 'mfb_timer_start')

    def timer_stop(self):
        ms = C.c_float()
        _lib.check(self._lib.mfb_timer_stop(self._h, C.byref(ms)), 'mfb_timer_stop')
        return ms.value

    def profile_enable(self, on=True):
        _lib.check(self._lib.mfb_profile_enable(self._h, int(bool(on))), 'mfb_profile_enable')

    def profile_read(self):
        cnt = (C.c_int * 2)()
        ms = (C.c_float * 2)()
        _lib.check(self._lib.mfb_profile_read(self._h, cnt, ms), 'mfb_profile_read')
        return (cnt[0], cnt[1]), (ms[0], ms[1])

    def sync(self):
        _lib.check(self._lib.mfb_sync(self._h), 'mfb_sync')


def analyze_filters(masks):
    """(start, length) of the common circular support window of the impulse responses of a filter bank
    complex64 [M, N] -- host-only, no GPU needed (length == N: no short support)."""
    lib = _lib.load()
    masks = np.ascontiguousarray(masks, dtype=np.complex64)
    st, ln = C.c_int(), C.c_int()
    _lib.check(lib.mfb_analyze_filters(_ptr(masks), masks.shape[0], masks.shape[1], C.byref(st), C.byref(ln)),
               'mfb_analyze_filters')
    return st.value, ln.value


def analyze_rank(masks):
    """Dimension of the span of a filter bank's impulse responses -- host-only, no GPU needed."""
    lib = _lib.load()
    masks = np.ascontiguousarray(masks, dtype=np.complex64)
    r = C.c_int()
    _lib.check(lib.mfb_analyze_rank(_ptr(masks), masks.shape[0], masks.shape[1], C.byref(r)), 'mfb_analyze_rank')
    return r.value


_xcorr_banks = {}


def customXCorr(a, b, N=None, device=0):
    """Drop-in for the reference's lib/customXCorr.customXCorr on the HIP transforms: N must be a power
    of two in [2^10, 2^22] (the soft combiner always pads to one, softCombiner.py:701-706)."""
    a = np.asarray(a)
    b = np.asarray(b)
    if N is None:
        N = max(len(a), len(b))
    N = int(N)
    if N & (N - 1) or not (1 << 10) <= N <= (1 << 22):
        raise ValueError(f'customXCorr on the GPU needs a power-of-two length in [2^10, 2^22], got {N}')
    key = (N, int(device))
    if key not in _xcorr_banks:
        _xcorr_banks[key] = MFBank(N.bit_length() - 1, 1, 1, device=device)
    return _xcorr_banks[key].xcorr(a, b)


def _as_bits_and_template(bits, template):
    b = np.asarray(bits)
    single = b.ndim == 1
    b2 = np.ascontiguousarray(b.reshape(1, -1) if single else b)
    if b2.dtype != np.uint8:
        if not np.all((b2 == 0) | (b2 == 1)):
            raise ValueError('expected a 0/1 bit stream')
        b2 = b2.astype(np.uint8)
    t = np.asarray(template)
    ti = np.ascontiguousarray(t, dtype=np.int8)
    if not np.array_equal(ti, t):
        raise ValueError('template must hold small integers (int8)')
    return single, b2, ti


def sync_find(bits, template, threshold, max_hits=1024, device=0):
    """Positions where the sync/preamble correlation reaches ``threshold`` and the scores there:
    ``np.where(np.convolve(bits, template) >= threshold)`` per stream, computed and thresholded on the
    GPU.  Returns (idx int32[], score int32[]) for a 1-D input, a list of such pairs for [B, L]."""
    lib = _lib.load()
    single, b2, ti = _as_bits_and_template(bits, template)
    B, L = b2.shape
    thr = int(np.ceil(threshold))
    while True:
        idx = np.empty((B, max_hits), dtype=np.int32)
        sc = np.empty((B, max_hits), dtype=np.int32)
        cnt = np.empty(B, dtype=np.int32)
        _lib.check(lib.mfb_sync_find(int(device), _ptr(b2), B, L, _ptr(ti), ti.size, thr, int(max_hits),
                                     _ptr(idx), _ptr(sc), _ptr(cnt)), 'mfb_sync_find')
        if cnt.max() <= max_hits:
            break
        max_hits = int(cnt.max())
    out = [(idx[b, :cnt[b]].copy(), sc[b, :cnt[b]].copy()) for b in range(B)]
    return out[0] if single else out


def sync_pinned_buffer(nbytes, device=0):
    """uint8 view of the library's page-locked staging buffer of this device (at least ``nbytes``): packed bit streams
    produced straight into it reach the device by plain asynchronous DMA.  One buffer per device: a later, larger request
    replaces it."""
    lib = _lib.load()
    p = C.c_void_p()
    _lib.check(lib.mfb_sync_pinned_buffer(int(device), C.c_size_t(int(nbytes)), C.byref(p)), 'mfb_sync_pinned_buffer')
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(int(nbytes),))


def sync_find_packed(packed, L, template, threshold, max_total=None, device=0, timing=False, flat=False):
    """``sync_find`` on PACKED bit streams: ``packed`` uint8 [B, row_bytes] (or [row_bytes]) in ``np.packbits`` layout, ``L``
    valid bits per stream, taps in {-1, 0, +1}.  XOR/AND + popcount on 64-bit windows on the GPU, exact; only the hits come
    back.  Returns a list of (idx int32[], score int32[]) per stream (one pair for a 1-D input) -- or, with ``flat``, the
    library's own (counts[B], idx[total], score[total]) --; with ``timing`` also the kernel time in ms."""
    lib = _lib.load()
    pk = np.asarray(packed)
    single = pk.ndim == 1
    pk = pk.reshape(1, -1) if single else pk
    if pk.dtype != np.uint8 or not pk.flags.c_contiguous:
        pk = np.ascontiguousarray(pk, dtype=np.uint8)
    B, row_bytes = pk.shape
    t = np.asarray(template)
    ti = np.ascontiguousarray(t, dtype=np.int8)
    if not np.array_equal(ti, t):
        raise ValueError('template must hold small integers (int8)')
    thr = int(np.ceil(threshold))
    max_total = int(max_total) if max_total else max(4096, 64 * B)
    while True:
        idx = np.empty(max_total, dtype=np.int32)
        sc = np.empty(max_total, dtype=np.int32)
        cnt = np.empty(B, dtype=np.int32)
        total = C.c_int32()
        ms = C.c_float()
        _lib.check(lib.mfb_sync_find_packed(int(device), _ptr(pk), B, int(L), row_bytes, _ptr(ti), ti.size, thr, max_total,
                                            _ptr(idx), _ptr(sc), _ptr(cnt), C.byref(total), C.byref(ms) if timing else None),
                   'mfb_sync_find_packed')
        if total.value <= max_total:
            break
        max_total = int(total.value)
    if flat:            # (counts per stream, positions, scores) as the library delivers them: stream b's hits start at
        res = (cnt, idx[:total.value], sc[:total.value])       # sum(counts[:b]); no per-stream Python objects
        return (res, ms.value) if timing else res
    off = np.concatenate(([0], np.cumsum(cnt)))
    out = [(idx[off[b]:off[b + 1]].copy(), sc[off[b]:off[b + 1]].copy()) for b in range(B)]
    res = out[0] if single else out
    return (res, ms.value) if timing else res


def sync_find_multi(bits, templates, thresholds, max_hits=256, device=0):
    """``sync_find`` for several templates against the same bit stream(s) in ONE library call (one host-device
    round trip): the decoder's header-mask and sync-flag searches of a block (reference decoder.py:96-113).
    Returns one ``sync_find`` result per template."""
    lib = _lib.load()
    b = np.asarray(bits)
    single = b.ndim == 1
    b2 = np.ascontiguousarray(b.reshape(1, -1) if single else b)
    if b2.dtype != np.uint8:
        u = b2.astype(np.uint8)
        if not np.array_equal(u, b2) or u.max(initial=0) > 1:
            raise ValueError('expected a 0/1 bit stream')
        b2 = u
    tis = []
    for t in templates:
        t = np.asarray(t)
        ti = np.ascontiguousarray(t, dtype=np.int8)
        if not np.array_equal(ti, t):
            raise ValueError('template must hold small integers (int8)')
        tis.append(ti)
    K = len(tis)
    packed = np.concatenate(tis)
    T = np.array([t.size for t in tis], dtype=np.int32)
    thr = np.array([int(np.ceil(x)) for x in thresholds], dtype=np.int32)
    if len(thr) != K:
        raise ValueError('one threshold per template')
    B, L = b2.shape
    while True:
        idx = np.empty((K, B, max_hits), dtype=np.int32)
        sc = np.empty((K, B, max_hits), dtype=np.int32)
        cnt = np.empty((K, B), dtype=np.int32)
        _lib.check(lib.mfb_sync_find_multi(int(device), _ptr(b2), B, L, _ptr(packed), _ptr(T), _ptr(thr), K, int(max_hits),
                                           _ptr(idx), _ptr(sc), _ptr(cnt)), 'mfb_sync_find_multi')
        if cnt.max() <= max_hits:
            break
        max_hits = int(cnt.max())
    out = []
    for k in range(K):
        per = [(idx[k, s, :cnt[k, s]].copy(), sc[k, s, :cnt[k, s]].copy()) for s in range(B)]
        out.append(per[0] if single else per)
    return out


class SyncFinder:
    """The decoder's sync search as an object (mfb_syncfinder_*): the templates it searches block after block (header mask,
    sync flag) and their thresholds are handed to the library once and stay on the device; ``begin(bits)`` copies a 1-D
    0/1 stream in and returns at once, ``end()`` waits and returns one (idx, score) pair per template; ``find`` does
    both.  uint8 streams are used as they are; other dtypes are converted without re-validation (the decoder's bits come
    from the demodulator's LUT)."""

    def __init__(self, templates, thresholds, max_hits=256, max_bits=1 << 16, device=0):
        self._lib = _lib.load()
        tis = []
        for t in templates:
            t = np.asarray(t)
            ti = np.ascontiguousarray(t, dtype=np.int8)
            if not np.array_equal(ti, t):
                raise ValueError('template must hold small integers (int8)')
            tis.append(ti)
        if len(tis) != len(thresholds):
            raise ValueError('one threshold per template')
        self.K = len(tis)
        self._templates = np.concatenate(tis)
        self._T = np.array([t.size for t in tis], dtype=np.int32)
        self._thr = np.array([int(np.ceil(x)) for x in thresholds], dtype=np.int32)
        self.device, self.max_bits = int(device), int(max_bits)
        self._h = C.c_void_p()
        self._pending = None
        self._create(int(max_hits))

    def _create(self, max_hits):
        self.close()
        self.max_hits = max_hits
        _lib.check(self._lib.mfb_syncfinder_create(C.byref(self._h), self.device, _ptr(self._templates), _ptr(self._T), _ptr(self._thr),
                                                   self.K, self.max_bits, max_hits), 'mfb_syncfinder_create')
        self._idx = np.empty((self.K, max_hits), dtype=np.int32)
        self._sc = np.empty((self.K, max_hits), dtype=np.int32)
        self._cnt = np.empty(self.K, dtype=np.int32)
        self._outs = (_ptr(self._cnt), _ptr(self._idx), _ptr(self._sc))

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            self._lib.mfb_syncfinder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def begin(self, bits):
        b = bits if (isinstance(bits, np.ndarray) and bits.dtype == np.uint8 and bits.flags.c_contiguous) else \
            np.ascontiguousarray(bits, dtype=np.uint8)
        _lib.check(self._lib.mfb_syncfinder_begin(self._h, _ptr(b), b.size), 'mfb_syncfinder_begin')
        self._pending = b

    def end(self):
        _lib.check(self._lib.mfb_syncfinder_end(self._h, *self._outs), 'mfb_syncfinder_end')
        bits, self._pending = self._pending, None
        if self._cnt.max() > self.max_hits:         # more hits than room: once more with enough of it
            self._create(int(self._cnt.max()))
            self.begin(bits)
            return self.end()
        return [(self._idx[k, :self._cnt[k]].copy(), self._sc[k, :self._cnt[k]].copy()) for k in range(self.K)]

    def find(self, bits):
        self.begin(bits)
        return self.end()


def sync_correlate(bits, template, device=0):
    """Batched full convolution of 0/1 bit streams with an integer template on the GPU
    (reference decoder.py:96,112 does this with np.convolve).  ``bits`` uint8 [L] or [B, L];
    returns int32 [L+T-1] or [B, L+T-1]."""
    lib = _lib.load()
    b = np.asarray(bits)
    single = b.ndim == 1
    b2 = np.ascontiguousarray(b.reshape(1, -1) if single else b)
    if b2.dtype != np.uint8:
        if not np.all((b2 == 0) | (b2 == 1)):
            raise ValueError('sync_correlate expects a 0/1 bit stream')
        b2 = b2.astype(np.uint8)
    t = np.asarray(template)
    ti = np.ascontiguousarray(t, dtype=np.int8)
    if not np.array_equal(ti, t):
        raise ValueError('template must hold small integers (int8)')
    B, L = b2.shape
    out = np.empty((B, L + ti.size - 1), dtype=np.int32)
    _lib.check(lib.mfb_sync_correlate(int(device), _ptr(b2), B, L, _ptr(ti), ti.size, _ptr(out)), 'mfb_sync_correlate')
    return out[0] if single else out
