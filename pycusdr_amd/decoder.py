"""Decoder: packet search in the demodulated bit stream by sync/preamble correlation.

Same surface as the reference's ``decoder.Decoder`` (reference decoder.py:16-293):
``Decoder(config, protocol)`` and ``findFrames(bits_raw, frameStartIdx) -> (packets, bits,
numSyncSig)``.  The two correlations that the reference computes with ``np.convolve``
(decoder.py:96,112) run on the GPU through ``mfb_sync_correlate`` (exact integers); the packet
state machine around them is host logic, as in the reference.  ``correlator`` may be injected (tests
pin the host logic on CPU-only machines with the oracle's correlator); the default is the HIP path
and raises if libmfbank.so is unavailable.
"""
import functools
import logging

import numpy as np

from .protocol import PacketEndDetect

log = logging.getLogger('pycusdr_amd.decoder')


def _hip_correlator(bits, template, device=0):
    from .mfbank import sync_correlate
    return sync_correlate(bits, template, device=device)


def _hip_finder(bits, template, threshold, device=0):
    from .mfbank import sync_find
    return sync_find(bits, template, threshold, device=device)


def _hip_finder_multi(bits, templates, thresholds, device=0):
    from .mfbank import sync_find_multi
    return sync_find_multi(bits, templates, thresholds, device=device)


def _config_device(config):
    """HIP device of this process: ``config['GPU'][<set>]['CUDA']['device']`` (the key the Demodulator reads,
    reference DB:178) of the first GPU set that names one; 0 otherwise."""
    try:
        for gpu_set in config.get('GPU', {}).values():
            dev = gpu_set.get('CUDA', {}).get('device')
            if dev is not None:
                return int(dev)
    except AttributeError:
        pass
    return 0


class Decoder:
    maxPacketLenBits = int(2 ** 13)
    minNumBitsBeforeProcessing = int(2 ** 10)

    def __init__(self, config, protocol, correlator=None, device=None):
        self.conf = config
        self.protocol = protocol
        # one process per GPU: the correlations run on this process's device, not on device 0
        self.device = _config_device(config) if device is None else int(device)
        # default: thresholded correlation on the GPU (positions + scores only come back); an injected
        # correlator returns the full np.convolve-style score array and is thresholded on the host
        # (a partial, not a lambda over self: a decoder must not sit in a reference cycle -- its device-side finder, page-locked
        # staging included, goes when the last reference does, not whenever the cycle collector next runs)
        self.correlate = correlator if correlator is not None else functools.partial(_hip_correlator, device=self.device)
        self._finder = _hip_finder if correlator is None else None
        self._multi = None
        self._flight = None
        self.preprocessor = protocol.decoderPreprocessor
        self.postprocessor = protocol.decoderPostprocessor
        self.mask = protocol.get_mask()
        self.syncSig = protocol.get_syncFlag()
        self.numBitsOverlap = protocol.numBitsOverlap
        self.bitsOverlapBuf = np.zeros(self.numBitsOverlap)
        self._prev = None              # (start of the last call's stream in the bit sequence, its hits): findFrames_batch
        self._abs_end = 0              # bits taken so far: position of the next block's first bit in the whole sequence
        self._edge_map = {}            # sequence position of a would-be stash start -> hits of its first T - 1 positions
        # cross-block packet state (FLAGS mode)
        self.headerFrameStartIdx = None
        self.packetBuffer = None
        self.headerMaskBitErrors = None
        self.packetEndDetectMode = protocol.packetEndDetectMode
        self.packetEndLenDecoder = protocol.packetEndLenDecoder
        self.packetSizes = protocol.packet_sizes
        self.packetLen = protocol.packetLen
        self.packetEndLenField = protocol.packetEndLenField
        self.packetEndLenFieldNumBytes = protocol.packetEndLenFieldNumBytes
        self.Packet = protocol.Packet

    # ------------------------------------------------------------------------------------------
    def hits(self, bits, template, threshold):
        """(positions, scores there) where the correlation with ``template`` reaches ``threshold`` --
        np.where(np.convolve(bits, template) >= threshold) of the reference (decoder.py:96-113)."""
        if self._finder is not None:
            return self._finder(bits, template, threshold, device=self.device)
        score = self.correlate(bits, template)
        idx = np.where(score >= threshold)[0]
        return idx, score[idx]

    def findFrames(self, bits_raw, frameStartIdx, debugMode=False):
        self.findFrames_begin(bits_raw, frameStartIdx)
        return self.findFrames_end()

    def close(self):
        """Release the device-side finder (stream, templates, page-locked staging) now; a later block creates it again."""
        if self._multi is not None:
            self._multi.close()
            self._multi = None
        self._flight = None

    def prepare(self):
        """Create the device-side finder now (page-locked staging, stream, templates on the device) instead of inside the
        first block of a stream.  No-op with an injected correlator."""
        if self._finder is not None and self._multi is None:
            from .mfbank import SyncFinder
            p = self.protocol
            self._multi = SyncFinder((self.mask, self.syncSig),
                                     (p.numOnesHeader - p.headerTol, p.numOnesSyncSig - p.syncSigTol), device=self.device)
            self._multi.find(np.zeros(max(len(self.mask), len(self.syncSig)) + 8, dtype=np.uint8))     # code objects loaded

    def findFrames_begin(self, bits_raw, frameStartIdx):
        """First half of ``findFrames``: preprocessing, the overlap stitch, and the two searches handed to the device.
        Returns at once; ``findFrames_end`` collects the hits and runs the packet state machine.  A caller may do other
        work in between (the streaming loop assembles and launches the next block), but must call _end before the next
        _begin: the state machine of block i decides what block i + 1 is stitched to (DEC:254-263)."""
        p = self.protocol
        bits_less_raw = self.preprocessor(bits_raw)
        self._abs_end += len(bits_less_raw)
        rawBits_DS = np.concatenate((self.bitsOverlapBuf, bits_less_raw))
        self.bitsOverlapBuf = rawBits_DS[-self.numBitsOverlap:]
        hits = None
        if self._finder is not None:
            # both searches of the block in one device round trip; templates and thresholds are handed over once (they are
            # read here, after get_mask / get_syncFlag have set the protocol's counts)
            self.prepare()
            self._multi.begin(rawBits_DS)
        else:
            hits = (self.hits(rawBits_DS, self.mask, p.numOnesHeader - p.headerTol),
                    self.hits(rawBits_DS, self.syncSig, p.numOnesSyncSig - p.syncSigTol))
        self._flight = (rawBits_DS, bits_less_raw, frameStartIdx, hits)

    def findFrames_end(self):
        rawBits_DS, bits_less_raw, frameStartIdx, hits = self._flight
        self._flight = None
        if hits is None:
            hits = self._multi.end()
        self._prev = None              # (a batch that follows searches its first block itself)
        return self._frames(rawBits_DS, bits_less_raw, frameStartIdx, hits)

    def _frames(self, rawBits_DS, bits_less_raw, frameStartIdx, hits):
        """The packet state machine of one call (DEC:101-293) on the stitched stream and the hits of its two searches."""
        (idxCand, candScore), (syncSigStartIdx, _) = hits
        back = len(self.mask) - 1                         # the peak sits on the template's last bit
        if isinstance(idxCand, list):                     # (findFrames_batch keeps its few hits per block in plain lists)
            if self.packetEndDetectMode == PacketEndDetect.FLAGS:
                idxCand, candScore, syncSigStartIdx = (np.asarray(idxCand, dtype=np.int64), np.asarray(candScore, dtype=np.int64),
                                                       np.asarray(syncSigStartIdx, dtype=np.int64))
                packetIdx = idxCand - back
            else:
                packetIdx = [i - back for i in idxCand]
        else:
            packetIdx = idxCand - back
        numSyncSig = len(syncSigStartIdx)

        if self.packetEndDetectMode == PacketEndDetect.FLAGS:
            packets = self._frames_by_flags(rawBits_DS, bits_less_raw, frameStartIdx, candScore, packetIdx, syncSigStartIdx)
        elif self.packetEndDetectMode == PacketEndDetect.FIXED:
            packets = self._frames_fixed(rawBits_DS, candScore, packetIdx)
        else:   # IN_DATA: the reference only calls protocol.packetDataProcessor() per header
            packets = []
            for _ in packetIdx:
                self.protocol.packetDataProcessor()
        return packets, bits_less_raw, numSyncSig

    # ---- several consecutive calls at once ---------------------------------------------------------
    def _search_streams(self, streams):
        """Both searches of every stream in ONE device round trip: [(mask hits, sync hits)] per stream."""
        p = self.protocol
        tmpls = (self.mask, self.syncSig)
        thr = (p.numOnesHeader - p.headerTol, p.numOnesSyncSig - p.syncSigTol)
        if self._finder is None:
            return [tuple(self.hits(s, t, h) for t, h in zip(tmpls, thr)) for s in streams]
        lens = [len(s) for s in streams]
        buf = np.zeros((len(streams), max(lens)), dtype=np.uint8)       # zero padding adds nothing to a full convolution
        for row, s in zip(buf, streams):
            row[:len(s)] = s
        res = _hip_finder_multi(buf, tmpls, thr, device=self.device)
        out = []
        for b, n in enumerate(lens):
            per = []
            for k, t in enumerate(tmpls):
                idx, sc = res[k][b]
                keep = idx < n + len(t) - 1                               # positions past this stream's own end: padding
                per.append((idx[keep], sc[keep]))
            out.append(tuple(per))
        return out

    def findFrames_batch(self, blocks_bits, frameStartIdx=0, ahead=None, edges=None):
        """``findFrames`` for several consecutive blocks: the same packets, returned bits and sync counts, call by call (a list
        of ``findFrames`` results), from ONE device round trip for the searches of all blocks instead of one per block -- or
        none: ``ahead[i]`` = the hits of block i's stream without a stash (the last ``numBitsOverlap`` bits before the block +
        its bits), as the batched block path delivers them with the block (None where it could not); ``edges[i]`` = for the first
        header hits of that stream, the hits among the first T - 1 positions of the stream a FIXED-mode decoder would restart at
        (20 bits in front of the header, DEC:254-263): ``(start relative to the stream, mask hits, sync hits)``.

        Every call's stream is a window [a, e) of the bit sequence so far: e the end of the block's bits, a either
        ``numBitsOverlap`` bits before the block (DEC:89-90) or -- FIXED mode with a packet still incomplete -- where the
        previous call stashed its candidate (DEC:254-263), which only the previous call's state machine can tell.  So the
        searches run ahead on the windows without a stash, and a block whose window turns out longer gets its hits put
        together, exactly: a full convolution's score at a position depends on the window only through which taps hang over
        its two ends, so positions at least T - 1 behind the default start are those of the default window, positions in front
        of that were positions of the previous call's stream (same bits, no overhang), and the first T - 1 positions are the
        previous call's when the start did not move -- else they are searched."""
        pre = [self.preprocessor(b) for b in blocks_bits]
        nb, nOv = len(pre), self.numBitsOverlap
        Ts = (len(self.mask), len(self.syncSig))
        if nOv < max(Ts) or len(self.bitsOverlapBuf) < nOv or (nb < 2 and ahead is None):
            return [self.findFrames(b, frameStartIdx) for b in blocks_bits]
        hist = np.concatenate([self.bitsOverlapBuf] + pre)
        ends = np.cumsum([len(self.bitsOverlapBuf)] + [len(b) for b in pre]).tolist()
        starts = [ends[i] - nOv for i in range(nb)]              # the windows without a stash, in hist's coordinates
        a, a_prev, prev = 0, None, None
        if self._prev is not None:                                # the previous call's window, in hist's coordinates (<= 0)
            a_prev, prev = self._prev[0] - self._seq_end + ends[0], self._prev[1]
        # a block has a handful of hits: plain lists from here on (a numpy call costs more than the whole list)
        ahead = [None if h is None else self._hit_lists(h) for h in ahead] if ahead is not None else [None] * nb
        base = self._abs_end - ends[0]                            # sequence position of hist[0]
        if edges is not None:
            if len(self._edge_map) > 64:
                self._edge_map = {k: v for k, v in self._edge_map.items() if k >= base}
            for i, cands in enumerate(edges):
                for a_rel, h0, h1 in cands or ():
                    self._edge_map[base + starts[i] + a_rel] = (h0, h1)
        self.ahead_blocks = getattr(self, 'ahead_blocks', 0) + sum(1 for h in ahead if h is not None)   # searches that came with the block
        # blocks whose run-ahead hits did not come with them: one device round trip for all of them.  The first block's real
        # stream is known: it is searched as it is when it cannot be put together from the previous call's hits
        first_direct = a != starts[0] and prev is None
        todo = [i for i in range(nb) if ahead[i] is None and not (i == 0 and first_direct)]
        streams = [hist[starts[i]:ends[i + 1]] for i in todo] + ([hist[a:ends[1]]] if first_direct else [])
        if streams:
            found = self._search_streams(streams)
            found = [self._hit_lists(h) for h in found]
            for i, h in zip(todo, found):
                ahead[i] = h
        out = []
        for i in range(nb):
            e, d = ends[i + 1], starts[i]
            stream = hist[a:e]
            if i == 0 and first_direct:
                hits = found[-1]
            elif a == d:
                hits = ahead[i]
            else:
                edge = None
                if a != a_prev:       # a new stash: its first T - 1 positions belong to no other window: delivered, or searched now
                    edge = self._edge_map.get(base + a)
                    if edge is None:
                        edge = self._search_streams([stream[:max(Ts) - 1]])[0]
                    edge = self._hit_lists(edge)
                hits = tuple(self._window_hits(Ts[k], prev[k], a_prev, a, d, ahead[i][k], edge[k] if edge else None) for k in range(2))
            self.bitsOverlapBuf = stream[-nOv:]
            out.append(self._frames(stream, pre[i], frameStartIdx, hits))
            a_prev, prev = a, hits
            a = e - len(self.bitsOverlapBuf)
        self._seq_end = ends[nb]
        self._abs_end += ends[nb] - ends[0]
        self._prev = (a_prev, prev)
        return out

    @staticmethod
    def _hit_lists(h):
        """((idx, score), (idx, score)) of the two templates as plain lists of ints."""
        if isinstance(h[0][0], list) and isinstance(h[1][0], list) and isinstance(h[0][1], list) and isinstance(h[1][1], list):
            return h
        return tuple((i if isinstance(i, list) else np.asarray(i).tolist(), s if isinstance(s, list) else np.asarray(s).tolist())
                     for i, s in h)

    @staticmethod
    def _window_hits(T, prev, a_prev, a, d, ahead, edge):
        """Hits of a T-tap template on the window [a, e), a < d, from the hits ``ahead`` of [d, e), ``prev`` of the previous call's
        window [a_prev, e_prev) (a_prev <= a, e_prev = d + numBitsOverlap) and -- when the start moved -- ``edge``, the hits of
        the window's first T - 1 bits searched on their own.  Lists in, lists out (index list, score list)."""
        ai, asc = ahead
        pi, ps = prev
        lim, shift = T - 1, d - a
        if edge is None:
            # the start did not move: every previous hit in front of the default window's own positions stays where it is (the
            # first T - 1 positions included); behind them the default window's hits, shifted
            cut = d - a_prev + lim
            oi, os_ = [], []
            for i, v in zip(pi, ps):
                if i < cut:
                    oi.append(i)
                    os_.append(v)
        else:
            oi, os_ = [], []
            for i, v in zip(edge[0], edge[1]):
                if i < lim:
                    oi.append(i)
                    os_.append(v)
            lo, hi = a + lim, d + lim                          # previous hits between the edge and the default window's own positions
            for i, v in zip(pi, ps):
                g = i + a_prev                                 # in the sequence's coordinates
                if lo <= g < hi:
                    oi.append(g - a)
                    os_.append(v)
        for i, v in zip(ai, asc):
            if i >= lim:
                oi.append(i + shift)
                os_.append(v)
        return oi, os_

    # ---- FIXED: packets of protocol.packetLen bits (reference decoder.py:245-280) ---------------
    def _frames_fixed(self, stream, candScore, packetIdx):
        packets = []
        for i, start in enumerate(packetIdx):
            if len(stream) - start < self.packetLen:
                # not all bits here yet: make sure the candidate survives in the overlap buffer
                keep_from = max((0, start - 20))
                if len(stream) - keep_from > self.numBitsOverlap:
                    self.bitsOverlapBuf = stream[keep_from:]
                break
            bits = stream[start:start + self.packetLen]
            if len(bits) > 0:
                packets.append(self.Packet(bits, start, self.protocol.numOnesHeader - candScore[i]))
            else:
                log.error('length of bits = 0. len(stream) = %d, idx start %d', len(stream), start)
        return packets

    # ---- FLAGS: a packet runs from a header to the next sync flag (reference decoder.py:122-243) --
    def _flag_end(self, syncSigStartIdx, after, strict_first):
        """Index where a frame starting before ``after`` ends, or None."""
        p = self.protocol
        if len(syncSigStartIdx) == 0:
            return None
        k = np.argmax(syncSigStartIdx > after)
        if strict_first and k == 0:
            return None
        if syncSigStartIdx[k] < p.numOnesSyncSig - p.syncSigTol:
            return None
        return np.min((syncSigStartIdx[k] + 16, syncSigStartIdx[-1]))

    def _frames_by_flags(self, stream, new_bits, frameStartIdx, candScore, packetIdx, syncSigStartIdx):
        p = self.protocol
        packets = []
        if self.headerFrameStartIdx is not None:
            end = self._flag_end(syncSigStartIdx, 0, strict_first=False)
            if end is None:
                room = self.maxPacketLenBits - len(self.packetBuffer)
                if room > len(new_bits):
                    self.packetBuffer = np.append(self.packetBuffer, new_bits)
                else:
                    # the reference's append result is discarded here (decoder.py:173)
                    packets.append(self.Packet(self.packetBuffer, self.headerFrameStartIdx, self.headerMaskBitErrors))
                    self.headerFrameStartIdx = None
            else:
                split = len(self.packetBuffer)
                self.packetBuffer = np.append(self.packetBuffer, stream[self.numBitsOverlap:end])
                packets.append(self.Packet(self.packetBuffer, self.headerFrameStartIdx, self.headerMaskBitErrors,
                                           frameSplitIdx=split))
                self.headerFrameStartIdx = None
        if self.headerFrameStartIdx is None:
            for i, start in enumerate(packetIdx):
                end = self._flag_end(syncSigStartIdx, start + 120, strict_first=True)
                errs = p.numOnesHeader - candScore[i]
                if end is None:
                    self.packetBuffer = stream[start:]
                    self.headerFrameStartIdx = frameStartIdx + start - self.numBitsOverlap
                    self.headerMaskBitErrors = errs
                else:
                    bits = stream[start:end]
                    if len(bits) >= 128:
                        packets.append(self.Packet(bits, start + frameStartIdx, errs))
        return packets
