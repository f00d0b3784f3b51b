"""Host driver of the MI355X receive hot path: the ``Demodulator`` class.

Same constructor, methods, return tuples and error behaviour as the reference's
``demodulator.demodulator_base.Demodulator`` (reference demodulator_base.py:36-1060), so that the
caller (``Demodulator_process.run``, reference demodulator_process.py:242-297) can switch to it
unchanged.  Everything the reference did with PyCUDA kernels and cuFFT goes through the C ABI of
libmfbank.so (hand-written HIP, include/mfbank.h); what the reference did in numpy on the host
(Doppler table, index interpolation, SNR, bit LUTs, block-overlap alignment, clipped-peak tagging)
stays numpy on the host.

There is no CPU fallback: constructing a Demodulator without libmfbank.so / a GPU raises.
"""
import logging
from enum import Enum

import numpy as np

from ..mfbank import MFBank

log = logging.getLogger('pycusdr_amd.demodulator')

SPEED_OF_LIGHT = 299792458.0  # scipy.constants.speed_of_light (reference DB:143)

# defaults of the block-overlap symbol check (reference DB:19-26)
SYMBOL_CHECK_OVERLAP_OFFSET = 20
SYMBOL_CHECK_ERROR_THRESHOLD = 1000
SYMBOL_CHECK_MATCH_NUM_ERRORS_ALLOWED = 10
SYMBOL_MISMATCHVAL = 0

TRUSTTYPE = np.int8  # reference __global__.py:25-26
DATATYPE = np.int8


class Operations(Enum):
    CENTRES_ABS = 0
    CENTRES_REAL = 1
    CENTRES_IMAG = 2


def doppler_bin_table(confRadio, rangeRateMax, Nfft):
    """Doppler search grid (reference DB:130-165): normalised bin positions, Hz lookup, integer
    spectrum shifts (negatives wrapped to the upper half) and the STX fixed shift."""
    baud, spsym = confRadio['baud'], confRadio['samplesPerSym']
    if_offset = confRadio['frequencyOffset_Hz']
    carrier = confRadio['frequency_Hz'] - if_offset
    centre = if_offset / baud / spsym
    stx_idx = np.int32(centre * Nfft)
    if stx_idx < 0:
        stx_idx += Nfft
    half_span = rangeRateMax * carrier / SPEED_OF_LIGHT / (baud * spsym)
    grid = np.linspace(centre - half_span, centre + half_span, confRadio['doppCarrierSteps'])
    noise_hz = confRadio.get('noise_measure_offset_Hz', False)
    if noise_hz:
        grid = np.concatenate((np.array([noise_hz / baud / spsym]), grid))
    shifts = np.round(grid * Nfft).astype(np.int32)
    shifts[shifts < 0] += Nfft
    return grid, grid * spsym * baud, shifts, int(stx_idx)


def _first_true(mask):
    """Index of the first True -- ``np.where(mask)[0][0]`` (reference DB:880-881) without building the index array; like it,
    IndexError when there is none."""
    i = int(np.argmax(mask)) if len(mask) else 0
    if not len(mask) or not mask[i]:
        raise IndexError('index 0 is out of bounds for axis 0 with size 0')
    return i


class Demodulator:
    """One receive channel: Doppler search + symbol demodulation of N-sample blocks."""
    backend = None

    def __init__(self, conf, protocol, radioName, shard=None):
        self.protocol = protocol
        self.radioName = radioName
        self.confRadio = confRadio = conf['Radios']['Rx'][radioName]
        self.confGPU = confGPU = conf['GPU'][confRadio['CUDA_settings']]
        self.shard = shard

        # block geometry
        self.sigLen = 2 ** confGPU['blockSize']
        self.sigOverlap = 2 ** confGPU['overlap']
        self.sigOverlapWin = int(self.sigOverlap / 2)
        self.Nfft = int(self.sigLen)
        self.clippedPeakSpan = confGPU['clippedPeakSpan']
        self.peakThresholdScale = confGPU['peakThresholdScale']
        self.disablePeakThresholding = confRadio.get('disablePeakThresholding', False)
        self.clippedPeakIPure = []
        self.clippedPeakI = []

        # block-overlap symbol check
        self.overlapOffset = confGPU.get('symbol_check_overlap_offset', SYMBOL_CHECK_OVERLAP_OFFSET)
        self.symbol_check_error_threshold = confGPU.get('symbol_check_error_threshold', SYMBOL_CHECK_ERROR_THRESHOLD)
        self.symbol_check_match_threshold = self.overlapOffset - confGPU.get(
            'symbol_check_match_num_errors_allowed', SYMBOL_CHECK_MATCH_NUM_ERRORS_ALLOWED)
        self.poswinP = []

        self.spsym = spsym = confRadio['samplesPerSym']
        self.spsymMin = int(spsym / 2)
        self.baudRate = confRadio['baud']
        self.sampleRate = self.baudRate * spsym
        self.voteWeight = confRadio.get('voteWeight', 1)

        self.windowWidth = confGPU['bitWindowWidth']
        self.windowWidthOffset = int(self.windowWidth / 2)
        self.CODE_SEARCH_MASK_OFFSET = 0
        self.SUM_ALL_MASKS_PYTHON = bool(getattr(protocol, 'SUM_ALL_MASKS_PYTHON', False))

        # Doppler grid
        self.num_dopplers = confRadio['doppCarrierSteps']
        self.centreFreqOffset = confRadio['frequencyOffset_Hz']
        self.doppIdxNorm, self.doppHzLUT, self.doppCyperSymNorm, self.doppOffsetIdx = doppler_bin_table(
            confRadio, conf['Radios']['rangeRateMax'], self.Nfft)
        self.doppIdxArrayLen = len(self.doppIdxNorm)
        self.doppIdxArrayOffset = self.doppIdxArrayLen - self.num_dopplers
        log.info('[%s]: Doppler scanning range %.0f to %.0f Hz', radioName, self.doppHzLUT[0], self.doppHzLUT[-1])

        cuda_cfg = confGPU.get('CUDA', {})
        self.numThreadsS = cuda_cfg.get('numThreadsS', 1024)
        if self.Nfft / self.numThreadsS != np.round(self.Nfft / self.numThreadsS):
            raise ValueError('[{}]: the size of the input signal has to be divisible by {}'.format(
                radioName, self.numThreadsS))
        device = cuda_cfg.get('device', 0)

        # filters and LUTs from the protocol plugin
        try:
            self.num_masks, masks = protocol.get_filter(self.Nfft, spsym, confGPU['xcorrMaskSize'])
        except Exception:
            log.error('[%s]: Exception occured in protocol %s while preparing filters', radioName, protocol.name)
            raise
        self._check_masks(masks)
        if self.num_masks > 32:
            log.warning('[%s]: more than 32 masks is not supported by the reference', radioName)
        try:
            self.bitLUT, self.symbolLUT = protocol.get_symbolLUT2(confGPU['xcorrMaskSize'])
        except Exception:
            log.error('[%s]: Exception occured in protocol %s while preparing symbol lookup table', radioName, protocol.name)
            raise
        if self.symbolLUT is not None:
            self.symbolLUT = np.asarray(self.symbolLUT)
        # a bit LUT of plain 0/1 values (every shipped protocol that has one) also as bytes: the lookup of a block's
        # symbols then yields the uint8 bits the caller gets anyway, without a float64 detour (same values)
        self._bitLUT_u8 = None
        if self.bitLUT is not None:
            lut = np.asarray(self.bitLUT)
            if lut.ndim == 1 and np.all((lut == 0) | (lut == 1)):
                self._bitLUT_u8 = lut.astype(np.uint8)

        # device side: this rank's slice of the Doppler bins (all of them without sharding).  The noise-reference bin
        # (doppIdxArrayOffset rows in front of the table, DB:148-159) is searched by every rank: the pick needs its score
        # (CU:550-554), and one more bin per rank costs less than a second exchange
        doff = self.doppIdxArrayOffset
        if shard is not None:
            self._bin_lo, self._bin_hi = shard.bin_range(self.num_dopplers)
        else:
            self._bin_lo, self._bin_hi = 0, self.num_dopplers
        nloc = self._bin_hi - self._bin_lo
        self.bank = MFBank(confGPU['blockSize'], nloc, self.num_masks,
                           window_width=self.windowWidth, sum_all_masks=self.SUM_ALL_MASKS_PYTHON,
                           code_search_mask_offset=self.CODE_SEARCH_MASK_OFFSET,
                           doppler_offset=doff, device=device)
        self.bank.set_filters(masks)
        self.bank.set_shifts(np.concatenate((self.doppCyperSymNorm[:doff],
                                             self.doppCyperSymNorm[doff + self._bin_lo:doff + self._bin_hi])))
        # optional search settings next to the reference's "CUDA" block (where it keeps batchSize / streams, DB:171-178):
        #   "HIP": {"search_path": "auto|segment|twopass", "search_basis": "filters|span", "search_mode": "transforms|energy"}
        hip_cfg = confGPU.get('HIP', {})
        if 'search_path' in hip_cfg:
            self.bank.set_search_path(hip_cfg['search_path'])
        if 'search_basis' in hip_cfg:
            self.bank.set_search_basis(hip_cfg['search_basis'])
        if 'search_mode' in hip_cfg:
            self.bank.set_search_mode(hip_cfg['search_mode'])
        if shard is not None:
            shard.attach(self.bank, self.num_dopplers, self.num_masks, sum_all=self.SUM_ALL_MASKS_PYTHON, noise_rows=doff)
        self._pick_bin = 0
        # one library call (one synchronisation) per block -- mfb_receive_block -- unless the Doppler bins are sharded over
        # ranks (the exchange sits between search and pick) or the config asks for the stage-by-stage calls:
        #   "HIP": {"one_call": false}
        self._one_call = bool(hip_cfg.get('one_call', True)) and shard is None and hasattr(self.bank, 'receive_block')
        self._pending = None
        if self._one_call:
            # room for the two spectrum windows computeSNR reads (DB:635-667), whichever neighbouring pair of bins the pick
            # falls between: the block call then always delivers them, and nothing ever has to be fetched from a spectrum
            # that a later block may already have replaced
            self.bank.BAND_CAPACITY, longest = self._snr_band_capacity(5)
            log.info('[%s]: SNR windows of up to %d spectrum bins travel with every block (longest window of this bin table: %d)',
                     radioName, self.bank.BAND_CAPACITY, longest)
            if longest > self.bank.BAND_CAPACITY:
                log.warning('[%s]: a pick between bins %d spectrum bins apart needs a longer SNR window than travels with the block: '
                            'it is fetched afterwards when no later block is on the device yet, else that block reports SNR = nan',
                            radioName, longest)

        # windowed argmax range of the symbol-rate estimate (reference DB:508-512)
        self.symsTolLow = 0.9 * spsym
        self.symsTolHigh = 1.1 * spsym
        self.codeRateAndPhaseOffsetLow = int(self.Nfft / self.symsTolLow)
        self.codeRateAndPhaseOffsetHigh = int(self.Nfft / self.symsTolHigh)
        self.dopplerIdxlast = 0
        log.info('[%s]: Initialization done', radioName)

    # ------------------------------------------------------------------------------------------
    def _check_masks(self, masks):
        """Shape / dtype validation of the protocol's filter bank (reference DB:252-257)."""
        if masks.shape != (self.num_masks, self.Nfft):
            raise ValueError('Masks provided by protocol {} expected to be of dimensions {}, got dimensions {}'.format(
                self.protocol.name, (self.num_masks, self.Nfft), masks.shape))
        if not isinstance(masks[0, 0], np.complex64):
            raise TypeError('Datatype of masks {}, expected {}'.format(type(masks[0, 0]), np.complex64))

    def close(self):
        bank = getattr(self, 'bank', None)
        if bank is not None:
            bank.close()
            self.bank = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def get_signalBufferHostPointer(self):
        """Writable page-locked complex64[N] the caller fills in place (reference DB:1055-1060)."""
        return self.bank.input

    # ---- input ---------------------------------------------------------------------------------
    def uploadToGPU(self, samples, device_ptr=None):
        """H2D copy + forward FFT (reference DB:548-558).  Sharded: rank 0 owns the IQ stream; its block is
        broadcast to every rank (``samples`` is ignored on the other ranks and may be None).  ``device_ptr``: the block
        already sits in device memory (N complex64 samples): no copy."""
        if device_ptr is not None:
            if self.shard is not None:
                raise ValueError('device-resident blocks are not supported together with Doppler-bin sharding')
            self.bank.upload_device(device_ptr)
            return
        if self.shard is None:
            self.bank.upload(samples)
            return
        sh = self.shard
        if sh.rank == 0:
            # straight from the library's page-locked input buffer into the shard's block buffer, on the shard's stream
            # (no temporary device tensor); the pick's read-back synchronises that stream before the caller gets the
            # buffer back
            raw = self.bank.input
            if samples is not raw and not (isinstance(samples, np.ndarray) and samples.ctypes.data == raw.ctypes.data
                                           and samples.size == raw.size):
                np.copyto(raw, np.asarray(samples, dtype=np.complex64))
            sh.broadcast_block(self.bank, sh.torch.from_numpy(raw.view(np.float32)))
        else:
            sh.broadcast_block(self.bank)

    def thresholdInput(self, samples):
        self._thresholdInput(samples)

    def _thresholdInput(self, samples):
        """Clip burst interference in place, twice, and remember where (reference DB:670-707)."""
        mag = np.abs(samples)
        thresh = self.peakThresholdScale * np.mean(mag)
        hot = np.where(mag > thresh)[0]
        samples[hot] = thresh * (samples[hot] / mag[hot])
        mag[hot] = np.abs(samples[hot])
        thresh = self.peakThresholdScale * np.mean(mag)
        hot = np.where(mag > thresh)[0]
        self.clippedPeakIPure = hot
        samples[hot] = thresh * (samples[hot] / mag[hot])
        if len(hot) > 0:
            gaps = np.diff(hot)
            real_gaps = np.where(gaps > 1)[0]
            self.peakMinGap = 100
            small = np.where(gaps[real_gaps] < self.peakMinGap)[0]
            marks = np.zeros(self.Nfft, dtype=np.int8)
            marks[hot] = 1
            for gi in real_gaps[small]:
                marks[hot[gi]:hot[gi] + gaps[gi]] = 1
            self.clippedPeakI = np.where(marks == 1)[0]
        else:
            self.clippedPeakI = hot.copy()

    def uploadAndFindUHF(self, samples):
        self._thresholdInput(samples)
        self.uploadToGPU(samples)
        return self._findUHF(samples)

    # ---- one call per block ----------------------------------------------------------------------
    def _receive_block(self, samples, device_ptr=None, fixed_shift=None):
        """Everything the device does with this block in one library call; the result waits in ``_pending`` for
        ``demodulateDevice``."""
        if device_ptr is not None:
            source = 'device'
        else:
            raw = self.bank.input
            if samples is not raw and not (isinstance(samples, np.ndarray) and samples.ctypes.data == raw.ctypes.data
                                           and samples.size == raw.size):
                np.copyto(raw, np.asarray(samples, dtype=np.complex64))
            source = 'pinned'
        self._pending = self.bank.receive_block(self.codeRateAndPhaseOffsetHigh,
                                                self.codeRateAndPhaseOffsetLow - self.codeRateAndPhaseOffsetHigh, self.spsymMin,
                                                op=Operations.CENTRES_ABS.value, snr_window=5, fixed_shift=fixed_shift,
                                                source=source, device_ptr=device_ptr)
        return self._pending

    def beginBlock(self, slot, source='pinned', fixed_shift=None, device_ptr=None):
        """Enqueue the whole device side of the block assembled in input buffer ``source`` ('pinned' / 'pinned2') and
        return at once; ``endBlock(slot)`` collects it.  Lets the caller run the sequential host stages of the previous
        block (and assemble the next one in the other buffer) while the device works -- the blocks themselves still
        execute, and are collected, strictly in order."""
        self.bank.begin_block(slot, self.codeRateAndPhaseOffsetHigh, self.codeRateAndPhaseOffsetLow - self.codeRateAndPhaseOffsetHigh,
                              self.spsymMin, op=Operations.CENTRES_ABS.value, snr_window=5, fixed_shift=fixed_shift, source=source,
                              device_ptr=device_ptr)

    def endBlock(self, slot):
        """(freqOffset_Hz, metric, clippedPeakIdx, SNR_dB) of the block begun in ``slot`` -- what ``uploadAndFindCarrier``
        returns -- with its symbol decisions waiting for ``demodulateDevice``."""
        self._pending = self.bank.end_block(slot)
        if self.backend == 'UHF':
            return self._estimate_from_block(self._pending)
        self.dopplerIdxlast = self.doppOffsetIdx
        return 0, 0, self.clippedPeakIPure, 0

    # ---- B consecutive blocks per call ----------------------------------------------------------
    def blockWindows(self, blocks_per_call):
        """The two page-locked sample windows of the batched path: each takes ``blocks_per_call`` consecutive blocks of the
        stream, block b at ``b * (N - overlap)`` (neighbours share their overlap samples)."""
        return self.bank.windows(int(blocks_per_call), self.sigLen - self.sigOverlap)

    def beginBlocks(self, slot, nblocks, source='window'):
        """``beginBlock`` for the first ``nblocks`` blocks of window ``source`` ('window' / 'window2'): the device side of all of
        them as one set of launches (mfb_receive_blocks_begin); ``endBlocks(slot)`` collects.  UHF back end, one-call path."""
        self.bank.begin_blocks(slot, nblocks, self.codeRateAndPhaseOffsetHigh,
                               self.codeRateAndPhaseOffsetLow - self.codeRateAndPhaseOffsetHigh, self.spsymMin,
                               op=Operations.CENTRES_ABS.value, snr_window=5, source=source)

    # ---- A12 / A13 / A14 of a batch on the device (stream_kernels.hpp) -----------------------------
    def enableStreamStages(self, decoder=None):
        """Let the device run the integer stages behind the symbol decisions for the blocks of a batch: the bit lookup
        (DB:1012-1051), the block-overlap alignment with its uint8 casts (DB:863-988, 859) and -- with ``decoder`` -- its two
        searches on the stream without a stash (DEC:89-113).  Returns False (and leaves them on the host) where a precondition
        does not hold: a bit LUT that is not plain 0 / 1, a decoder preprocessor that is not the identity, templates with
        other taps than -1 / 0 / +1, more overlap bits than the device keeps.  Results are the host code's, bit for bit; blocks
        the device flags as irregular go through the host code."""
        from ..protocol.protocolBase import ProtocolBase
        self._stages = False
        if not self._one_call or self.backend != 'UHF' or self.overlapOffset + 1 > 32:
            return False
        kw = {}
        if self._bitLUT_u8 is not None and len(self._bitLUT_u8) <= 256:
            kw['bit_lut'] = self._bitLUT_u8
        elif (self._bitLUT_u8 is None and self.bitLUT is None and self.symbolLUT is not None and len(self.symbolLUT.shape) == 3 and
              self.symbolLUT.shape[1] == 2 and self.symbolLUT.size <= 2048):
            kw['nrzs_lut'] = self.symbolLUT
        else:
            return False
        self._stage_decoder = None
        if decoder is not None and getattr(decoder, '_finder', None) is not None:
            p = decoder.protocol
            ok = (getattr(type(p), 'decoderPreprocessor', None) is ProtocolBase.decoderPreprocessor and decoder.numBitsOverlap <= 4096 and
                  decoder.numBitsOverlap >= max(len(decoder.mask), len(decoder.syncSig)) and
                  all(np.array_equal(np.asarray(t), np.asarray(t, dtype=np.int8)) and np.abs(np.asarray(t)).max() <= 1
                      for t in (decoder.mask, decoder.syncSig)))
            if ok:
                kw.update(templates=(decoder.mask, decoder.syncSig), bits_overlap=decoder.numBitsOverlap,
                          thresholds=(p.numOnesHeader - p.headerTol, p.numOnesSyncSig - p.syncSigTol))
                self._stage_decoder = decoder
        self.bank.set_stream_stages(self.sigOverlap, self.overlapOffset, self.symbol_check_match_threshold,
                                    self.symbol_check_error_threshold, **kw)
        self._stages = True
        self._stream_dirty = True        # nothing is known on the device until seedStreamStages
        return True

    def seedStreamStages(self):
        """Hand the device the state the next batch starts from -- this object's alignment state (``poswinP``, ``posSymEnd``) and
        the decoder's last ``numBitsOverlap`` bits.  Nothing may be in flight."""
        post = np.asarray(self.poswinP, dtype=np.uint8)
        end = np.asarray(getattr(self, 'posSymEnd', []), dtype=np.uint8)
        ring = None
        if self._stage_decoder is not None:
            ring = np.asarray(self._stage_decoder.bitsOverlapBuf[-self._stage_decoder.numBitsOverlap:], dtype=np.uint8)
        if len(post) > 512 or len(end) > 32:
            return False
        self.bank.stream_seed(post, end, ring)
        self._dev_tail = (self.poswinP, getattr(self, 'posSymEnd', None))      # what the device assumes in front of the next block
        self._stream_dirty = False
        return True

    def waitBlocks(self, slot):
        """Wait for the batch begun in ``slot`` and take its records off the device (``endBlocks(slot, record)`` turns them into
        the per-block results later: the receive loop queues the next window's copies in between)."""
        return self.bank.end_blocks_record(slot)

    def endBlocks(self, slot, record=None):
        """One ``((freqOffset_Hz, metric, clippedPeakIdx, SNR_dB), device record)`` per block of the batch begun in ``slot``,
        in stream order: what ``endBlock`` + ``demodulateDevice`` return block by block."""
        R = self.bank.end_blocks_record(slot) if record is None else record
        s, nb = R.s, R.nb
        snr = self._batch_snr(R)
        stages = R.stages and getattr(self, '_stages', False)
        nrzs = self._bitLUT_u8 is None
        out = []
        prev_export = getattr(self, '_dev_tail', None)
        empty = np.zeros(0, dtype=np.int64)
        # (per-block Python is what bounds the loop with the decoder in the thread: the columns once, the object's "last block"
        # attributes once behind the loop)
        pick_valid, low, high, frac, picks, counts = s['pick_valid'], s['low'], s['high'], s['frac'], s['pick'], s['count']
        hz, hz_off, clipped = self.doppHzLUT, self.centreFreqOffset, self.clippedPeakIPure
        for b in range(nb):
            if not pick_valid[b]:       # NaN index (all-zero block): skip the block (reference DB:625-630)
                log.error('Error occurred during find_UHF -- skipping block. Message: cannot convert float NaN to integer')
                est = (0., 0., clipped, 0.)
            else:
                lowIdx, highIdx = low[b], high[b]
                lowVal, highVal = hz[lowIdx], hz[highIdx]
                bestDopplerScaled = lowVal + (highVal - lowVal) * frac[b]
                if snr is not None:
                    SNR = snr[b]
                else:
                    l0, l1 = s['band_len'][b]
                    if R.bands is not None and l0 <= R.bcap and l1 <= R.bcap:
                        SNR = self.computeSNR(lowIdx, highIdx, 5, bands=(R.bands[b, 0, :l0], R.bands[b, 1, :l1]))
                    else:
                        # a batch's spectra live in the batch workspace, never in the handle's one-block spectrum: windows
                        # beyond the record's capacity (capped at 2^16 elements, _snr_band_capacity) cannot be fetched later
                        log.error('[%s]: the SNR windows of this block (%d + %d elements) exceed the %d delivered with a batch: SNR = nan',
                                  self.radioName, l0, l1, R.bcap)
                        SNR = float('nan')
                est = (bestDopplerScaled - hz_off, float(picks[b][1]) / self.Nfft * self.sampleRate, clipped, SNR)
            n = counts[b]
            if s['rate_fallback'][b]:
                log.error('Code rate result 0 should not happen but happened -- fixing it to 10')
            rec = {'spSym': s['spSym'][b], 'symbols': R.sym[b, :n], 'centres': R.cen[b, :n],
                   'trust': R.mag[b].view(TRUSTTYPE)[:n], 'clipped': empty}
            if stages:
                if s['a13_status'][b]:
                    nw = s['a13_nwin'][b]
                    post, end = R.post[b, :s['a13_npost'][b]], R.end[b, :s['a13_nend'][b]]
                    if nrzs:
                        post, end = post.view(np.bool_), end.view(np.bool_)
                    # (the kept bits / centres mod 256 / trust bytes, the block's own tail, the tail the device assumed in front of it)
                    # (status 2: numpy could not form the alignment's first comparison -- the reference's log line goes out with the block)
                    raised = (min(self.overlapOffset, s['a13_prev_npost'][b]), min(self.overlapOffset, nw)) if s['a13_status'][b] == 2 else None
                    rec['_a13'] = (R.bits[b, :nw], R.cen8[b, :nw], R.trust[b, :nw], post, end, prev_export, raised)
                    prev_export = (post, end)
                else:
                    prev_export = None
                if R.templates == 2 and s['sync_valid'][b] and max(s['sync_count'][b]) <= R.max_hits:
                    c0, c1 = s['sync_count'][b]
                    # (plain lists: a block has a handful of hits, and the decoder composes them element by element)
                    hb = R.hits[b]
                    rec['_sync'] = ((hb[0, 0, :c0].tolist(), hb[0, 1, :c0].tolist()), (hb[1, 0, :c1].tolist(), hb[1, 1, :c1].tolist()))
                    if c0 and R.edges is not None:
                        # the leading positions of the streams the decoder would restart at for the first header hits
                        eh, cands = R.edge_hits, []
                        for E in R.edges[b].tolist():
                            if E[1]:
                                n0, n1 = E[2], E[3]
                                cands.append((E[0], (E[4:4 + n0], E[4 + 2 * eh:4 + 2 * eh + n0]), (E[4 + eh:4 + eh + n1], E[4 + 3 * eh:4 + 3 * eh + n1])))
                        rec['_edges'] = cands
            out.append((est, rec))
        if nb:
            # what the object remembers of its last block (DB:612-632, 730-752)
            last = nb - 1
            if pick_valid[last]:
                self._pick_bin = low[last]
                self.dopplerIdxlast = np.int32(s['shift'][last])
            else:
                self.dopplerIdxlast = 0
            self._codeRateResult = np.array(s['cr'][last], dtype=np.float32)
            self.magnitudes = R.mag[last, :counts[last]]
        if stages:
            self._dev_tail = prev_export
        return out

    def _batch_snr(self, R):
        """computeSNR (DB:635-667) of every block of a batch in one go when their windows have the same two lengths (the usual
        case: the pick sits between the same pair of bins) -- the same float32 operations element by element, so the same
        values; None otherwise (the blocks are then done one by one)."""
        s = R.s
        if R.bands is None or not R.searched or not all(s['pick_valid']):
            return None
        lens = s['band_len']
        l0, l1 = lens[0]
        if any(x != lens[0] for x in lens) or l0 > R.bcap or l1 > R.bcap:
            return None
        with np.errstate(divide='ignore', invalid='ignore'):
            sig = np.abs(R.bands[:, 0, :l0]).mean(axis=1) if l0 else np.full(R.nb, np.nan, np.float32)
            noise = np.abs(R.bands[:, 1, :l1]).mean(axis=1) if l1 else np.full(R.nb, np.nan, np.float32)
            return list(20 * np.log10(sig / noise - 1))

    def _estimate_from_block(self, blk):
        """The host half of __findUHF (reference DB:604-632) on what mfb_receive_block returned: Hz interpolation, SNR,
        the tuple the caller gets.  The shift interpolation itself ran on the device, same float64 operations."""
        best = blk['pick']
        if not blk['pick_valid']:       # NaN index (all-zero block): skip the block (reference DB:625-630)
            log.error('Error occurred during find_UHF -- skipping block. Message: cannot convert float NaN to integer')
            self.dopplerIdxlast = 0
            return 0., 0., self.clippedPeakIPure, 0.
        lowIdx, highIdx, frac = blk['low'], blk['high'], blk['frac']
        self._pick_bin = lowIdx
        lowVal, highVal = self.doppHzLUT[lowIdx], self.doppHzLUT[highIdx]
        bestDopplerScaled = lowVal + (highVal - lowVal) * frac
        self.dopplerIdxlast = np.int32(blk['shift'])
        SNR = self.computeSNR(lowIdx, highIdx, 5, bands=blk['bands'])
        freqOffset = bestDopplerScaled - self.centreFreqOffset
        sdev_Hz = float(best[1]) / self.Nfft * self.sampleRate
        return freqOffset, sdev_Hz, self.clippedPeakIPure, SNR

    # ---- Doppler search ------------------------------------------------------------------------
    def _device_search(self):
        """Device part of the search: [index, metric] as float32."""
        if self.shard is None:
            return self.bank.find_carrier()
        return self.shard.search_and_pick(self.bank, self._bin_lo)

    def _findUHF(self, samples=None):
        """Doppler estimate of the uploaded block (reference DB:567-632).  Returns
        (freqOffset_Hz, metric, clippedPeakIdx, SNR_dB)."""
        best = np.empty(2, dtype=np.float32)
        best[0], best[1] = self._device_search()
        try:
            lowIdx = int(best[0])
            self._pick_bin = lowIdx
            highIdx = int(np.ceil(best[0]))
            frac = float(best[0]) % 1
            lowVal, highVal = self.doppHzLUT[lowIdx], self.doppHzLUT[highIdx]
            bestDopplerScaled = lowVal + (highVal - lowVal) * frac
            s_lo, s_hi = int(self.doppCyperSymNorm[lowIdx]), int(self.doppCyperSymNorm[highIdx])
            self.dopplerIdxlast = np.int32(np.round(s_lo + (s_hi - s_lo) * frac))
            SNR = self.computeSNR(lowIdx, highIdx, 5)
            freqOffset = bestDopplerScaled - self.centreFreqOffset
            # the reference names this 'sdev_Hz' (DB:623): metric scaled as if it were an index spread
            sdev_Hz = float(best[1]) / self.Nfft * self.sampleRate
        except ValueError as e:   # NaN index (all-zero block): skip the block (reference DB:625-630)
            log.error('Error occurred during find_UHF -- skipping block. Message: %s', e)
            self.dopplerIdxlast = 0
            freqOffset = 0.
            sdev_Hz = 0.
            SNR = 0.
        return freqOffset, sdev_Hz, self.clippedPeakIPure, SNR

    def _snr_band_capacity(self, windowWidth):
        """(capacity, longest): the longest spectrum window ``computeSNR`` can ask for (reference DB:635-667), over every
        (low, high) the pick can produce -- high = low or low + 1 --, and the capacity reserved for it in every block's result
        record: rounded up to a power of two, at least 256 elements and at most 2^16 (two windows of 2^16 complex64 are 1 MiB
        of page-locked staging and read-back per block; a sparse table over a wide range -- one neighbour gap or a wrap pair of
        half a spectrum -- must not make every block ship 16 MiB)."""
        # (the noise-reference rows in front of the table never take part in a pick: its index starts behind them, CU:536)
        N, sh = self.Nfft, self.doppCyperSymNorm[self.doppIdxArrayOffset:].astype(np.int64)
        lo = np.concatenate((sh, sh[:-1]))
        hi = np.concatenate((sh, sh[1:]))

        def lengths(a, b):
            # X[a-w:b+w], or X[a-w:] + X[:b+w] when the band wraps (a > b), with numpy's slice semantics
            def sl(start, stop):
                start = np.where(start < 0, np.maximum(start + N, 0), np.minimum(start, N))
                stop = np.where(stop < 0, np.maximum(stop + N, 0), np.minimum(stop, N))
                return np.maximum(stop - start, 0)
            w = windowWidth
            return np.where(a > b, sl(a - w, np.full_like(a, N)) + sl(np.zeros_like(b), b + w), sl(a - w, b + w))
        longest = int(max(lengths(lo, hi).max(), lengths((lo + N // 2) % N, (hi + N // 2) % N).max(), 1))
        return int(min(N, 1 << 16, max(256, 1 << int(np.ceil(np.log2(longest)))))), longest

    def _spectrum_slice(self, a, b):
        """``X[a:b]`` with numpy slice semantics, fetching only that window from the device."""
        start, stop, _ = slice(a, b).indices(self.Nfft)
        if stop <= start:
            return np.empty(0, dtype=np.complex64)
        return self.bank.get_spectrum(start, stop - start)

    def computeSNR(self, doppMatchLow, doppMatchHigh, windowWidth, bands=None):
        """Signal band vs the band half a spectrum away (reference DB:635-667).  ``bands``: the two spectrum windows, if
        the block call delivered them (the same elements, in the same order, as the slices below)."""
        if bands is not None:
            with np.errstate(divide='ignore', invalid='ignore'):
                return 20 * np.log10(np.mean(np.abs(bands[0])) / np.mean(np.abs(bands[1])) - 1)
        if getattr(self.bank, 'flights', 0) > 0:
            # a later block is on the device already: its spectrum has replaced (or is replacing) this block's
            log.error('[%s]: the SNR windows of this block were not delivered with it and its spectrum is gone: SNR = nan', self.radioName)
            return float('nan')
        lo = int(self.doppCyperSymNorm[doppMatchLow])
        hi = int(self.doppCyperSymNorm[doppMatchHigh])
        nlo = (lo + int(self.Nfft // 2)) % self.Nfft
        nhi = (hi + int(self.Nfft // 2)) % self.Nfft

        def band(a, b):
            if a > b:   # band wraps around 0 Hz
                return np.mean(np.concatenate((np.abs(self._spectrum_slice(a - windowWidth, None)),
                                               np.abs(self._spectrum_slice(None, b + windowWidth)))))
            return np.mean(np.abs(self._spectrum_slice(a - windowWidth, b + windowWidth)))
        with np.errstate(divide='ignore', invalid='ignore'):
            return 20 * np.log10(band(lo, hi) / band(nlo, nhi) - 1)

    # ---- demodulation ----------------------------------------------------------------------------
    def findCodeRateAndPhaseGPU(self):
        """Matched filters at ``dopplerIdxlast`` + symbol rate / phase (reference DB:711-752 together
        with the shift-multiply and inverse FFT of DB:776-785, which the C ABI fuses into one call)."""
        res = np.empty(3, dtype=np.float32)
        res[0], res[1], res[2] = self.bank.demodulate(
            int(self.dopplerIdxlast), self.codeRateAndPhaseOffsetHigh,
            self.codeRateAndPhaseOffsetLow - self.codeRateAndPhaseOffsetHigh)
        self._codeRateResult = res
        k = float(res[0])              # host arithmetic is float64, as numpy<2 promoted it
        if k == 0.0:
            log.error('Code rate result 0 should not happen but happened -- fixing it to 10')
            spSym = 10
        else:
            spSym = self.Nfft / k
        codeOffset = -float(res[1]) / np.pi * spSym / 2
        if codeOffset < 0:
            codeOffset += spSym - 1
        return spSym, codeOffset

    def demodulateUHF(self):
        return self._demodulate()

    def demodulateSTX(self):
        self.dopplerIdxlast = self.doppOffsetIdx
        return self._demodulate()

    def _demodulate(self):
        """Symbols of the uploaded block at the found shift (reference DB:765-859).  Returns
        (bits uint8[], centres uint8[] (mod 256), trust uint8[], spSym)."""
        # (Doppler-sharded: every rank runs this stage -- each holds the block, the filters and the pick -- so the
        # symbol-overlap state below and the caller's decoder see one contiguous stream whichever rank owns the picked bin)
        return self.demodulateHost(self.demodulateDevice())

    def demodulateDevice(self):
        """The device half of the demodulation (reference DB:776-785, 711-752, 991-1009): matched filters at
        ``dopplerIdxlast``, symbol rate and phase, per-symbol decisions.  Nothing here depends on earlier blocks, so any
        process may run it for any block (time-chunk sharding, dist.BlockShard); the result travels as plain arrays."""
        blk, self._pending = self._pending, None
        if blk is not None and blk['shift'] == int(self.dopplerIdxlast):
            # the block call already ran this stage at the shift it found (same kernels, same float64 arithmetic)
            if blk['rate_fallback']:
                log.error('Code rate result 0 should not happen but happened -- fixing it to 10')
            self._codeRateResult = np.array(blk['cr'], dtype=np.float32)
            self.magnitudes = blk['magnitudes']
            spSym = blk['spSym']
            idxSymbol, centres = blk['symbols'], blk['centres']
            trustSymbol = self.magnitudes.view(TRUSTTYPE)[:len(idxSymbol)].copy()
        else:
            spSym, codeOffset = self.findCodeRateAndPhaseGPU()
            idxSymbol, _, centres, _, _, trustSymbol = self.cudaFindCentres(spSym, codeOffset, Operations.CENTRES_ABS)
        return {'spSym': spSym, 'symbols': idxSymbol, 'centres': centres, 'trust': trustSymbol,
                'clipped': np.asarray(self.clippedPeakIPure, dtype=np.int64)}

    def hostBits(self, rec):
        """A12 on a block's device record: (bits of every symbol, number of impossible NRZ-S transitions).  Depends on this block
        alone; kept in the record so that the tail and the host stage share it."""
        if '_bits' not in rec:
            idxSymbol, centres = rec['symbols'], rec['centres']
            if self._bitLUT_u8 is not None:
                rec['_bits'] = (self._bitLUT_u8[idxSymbol], 0)
            else:
                dataBits, symError_t = self.extractBits(centres, idxSymbol)
                rec['_bits'] = (dataBits, len(symError_t))
        return rec['_bits']

    def overlapTail(self, rec):
        """What the NEXT block's alignment needs of this one (reference DB:977-979): the bits behind this block's window
        (``poswinP``) and the last ``overlapOffset + 1`` bits inside it (``posSymEnd``).  Both are functions of this block's
        device results alone -- the +-1 adjustment of the window's START (DB:938-957) cannot reach its last bits -- so
        the owner of block i can hand them to the owner of block i + 1 without waiting for block i - 1 (time-chunk sharding,
        dist.BlockShard).  'exact' is False for windows too short for that argument (fewer than overlapOffset + 2 symbols)."""
        dataBits, _ = self.hostBits(rec)
        centres = rec['centres']
        start = _first_true(centres >= self.sigOverlapWin)
        end = _first_true(centres > (self.Nfft - self.sigOverlapWin))
        o = self.overlapOffset
        return {'post': dataBits[end:], 'end': dataBits[start:end][-o - 1:], 'exact': bool(end - start >= o + 2)}

    def demodulateHost(self, rec, prev_tail=None):
        """The sequential half (reference DB:1012-1051, 863-988, 817-859): bit lookup, alignment against the previous
        block (stateful: ``poswinP``, ``posSymEnd``), clipped-peak tagging, uint8 casts.  Must see the blocks in order --
        or be given the previous block's ``overlapTail`` as ``prev_tail`` (then any process may run any block)."""
        a13 = rec.get('_a13')
        if a13 is not None and prev_tail is None and not getattr(self, '_stream_dirty', True) and not len(rec['clipped']):
            # the device ran A12 / A13 for this block (stream_kernels.hpp) in front of the tail a13[5]; that must be THIS object's
            # state -- the arrays the previous device block left here, or equal ones
            bits, cen8, trust8, post, end, assumed, raised = a13
            mine = (self.poswinP, getattr(self, 'posSymEnd', None))
            if assumed is not None and ((mine[0] is assumed[0] and mine[1] is assumed[1]) or
                                        (mine[1] is not None and assumed[1] is not None and np.array_equal(mine[0], assumed[0])
                                         and np.array_equal(mine[1], assumed[1]))):
                if raised is not None:      # the reference logs the failed comparison and carries on (DB:965-967)
                    log.error('symbol overlap failed. reason: %s',
                              'operands could not be broadcast together with shapes (%d,) (%d,) ' % raised)
                self.poswinP, self.posSymEnd = post, end
                self.stage_blocks = getattr(self, 'stage_blocks', 0) + 1        # blocks whose A12 / A13 the device did
                # (copies: what the caller keeps must not keep the whole batch's record alive)
                return bits.copy(), cen8.copy(), trust8.copy(), rec['spSym']
        if getattr(self, '_stages', False):
            self._stream_dirty = True        # this block goes through the host code: the device's chain is broken until it is seeded again
        spSym, idxSymbol, centres, trustSymbol = rec['spSym'], rec['symbols'], rec['centres'], rec['trust']
        dataBits, noError = self.hostBits(rec)
        if prev_tail is not None:
            self.poswinP, self.posSymEnd = prev_tail['post'], prev_tail['end']
        centresWin, dataBitsWin, trustSymbolWin, _ = self.checkSymbolOverlap(noError, centres, idxSymbol, dataBits, trustSymbol)

        # tag symbols next to clipped interference peaks (reference DB:830-837)
        if len(rec['clipped']):
            marks = np.zeros(self.Nfft, dtype=bool)
            spSymc = int(np.ceil(spSym))
            for cp in rec['clipped']:
                marks[cp - 2 * spSymc:cp + 2 * spSymc + 1] = 1
            trustSymbolWin[marks[centresWin]] = -2
        return dataBitsWin.astype(np.uint8), centresWin.astype(np.uint8), trustSymbolWin.astype(np.uint8), spSym

    def cudaFindCentres(self, spSym, codePhase, operation=Operations.CENTRES_ABS):
        """Per-symbol argmax over filters and a W-sample window (reference DB:991-1009).

        Quirk kept on purpose: the reference allocates and reads back the magnitude buffer as int8
        although the kernel writes float32 (DB:472,1005-1006), so ``trust`` is the raw little-endian
        bytes of the first S/4 magnitudes."""
        if spSym < self.spsymMin:
            spSym = self.spsymMin
        count = int(self.Nfft / spSym)
        symbols, centres, mag = self.bank.find_centres(np.float32(spSym), np.float32(codePhase), operation.value, count)
        self.magnitudes = mag
        trust = mag.view(TRUSTTYPE)[:count].copy()
        return symbols, [], centres, 0, 0, trust

    findCentres = cudaFindCentres

    def extractBits(self, centres, symbols):
        """Filter index -> bit (reference DB:1012-1023)."""
        if self.bitLUT is None:
            if self.symbolLUT is not None and len(self.symbolLUT.shape) == 3:
                return self.extractBitsNRZs(centres, symbols)
            raise NotImplementedError('protocol provides neither a bitLUT nor a 3-D NRZ-S symbolLUT '
                                      '(the reference calls an undefined extractBitsOld here, DB:1017)')
        return self.bitLUT[symbols], []

    def extractBitsNRZs(self, centresCoherent, symbols):
        """NRZ-S transition decode with the 3-D LUT symbolLUT[sym][0|1][successors]
        (reference DB:1026-1051).  Returns (bits bool[S-1], indices of impossible transitions)."""
        nxt = symbols[1:, None]
        is_one = np.any(nxt == self.symbolLUT[symbols[:-1], 0, :], axis=1)
        is_zero = np.any(nxt == self.symbolLUT[symbols[:-1], 1, :], axis=1)
        bad = np.where((is_one + is_zero) == 0)[0].tolist()
        is_one[bad] = int(SYMBOL_MISMATCHVAL)
        return is_one, bad

    def checkSymbolOverlap(self, noError, centres, idxSymbol, dataBits, trustSymbol):
        """Keep the symbols whose centre lies in [ov/2, N-ov/2] and repair a +-1 symbol slip against
        the previous block (reference DB:863-988).  Stateful: ``poswinP`` (symbols after this
        block's window) and ``posSymEnd`` (last offset+1 kept symbols)."""
        start = _first_true(centres >= self.sigOverlapWin)
        end = _first_true(centres > (self.Nfft - self.sigOverlapWin))
        win = dataBits[start:end]
        pre = dataBits[:start]
        o = self.overlapOffset
        thr = self.symbol_check_match_threshold
        try:
            if noError > self.symbol_check_error_threshold:
                pass
            elif len(self.poswinP) > 0:
                prev_post, prev_end = self.poswinP, self.posSymEnd
                aligned = np.all(prev_post[:o] == win[:o]) or np.all(prev_end[-o:] == pre[-o:])
                if not aligned:
                    m_pre = (np.sum(prev_post[:o] == win[:o]),             # as is
                             np.sum(prev_post[:o] == win[1:o + 1]),        # this block starts a symbol early
                             np.sum(prev_post[1:o + 1] == win[0:o]))       # ... a symbol late
                    m_pos = (np.sum(prev_end[-o:] == pre[-o:]),
                             np.sum(prev_end[-o - 1:-1] == pre[-o:]),
                             np.sum(prev_end[-o:] == pre[-o - 1:-1]))
                    if thr < m_pre[1] and m_pre[1] == np.max(m_pre):
                        if thr < m_pos[1] and m_pos[1] == np.max(m_pos):
                            start += 1
                            win = win[1:]
                    elif thr < m_pre[2] and m_pre[2] == np.max(m_pre):
                        if thr < m_pos[2] and m_pos[2] == np.max(m_pos):
                            start -= 1
                            win = np.r_[pre[-1], win]
        except Exception as e:   # the reference logs and carries on (DB:965-967)
            log.error('symbol overlap failed. reason: %s', e)

        dataBitsWin = dataBits[start:end]
        self.poswinP = dataBits[end:]
        self.posSymEnd = dataBitsWin[-o - 1:]
        return centres[start:end], dataBitsWin, trustSymbol[start:end], win
