"""Fixed-size blocks out of arbitrary-size chunks: the buffer half of the reference's sigFIFO.py.

Three pieces:

* ``BlockAssembler`` -- what the streaming loop uses.  Chunks of any size are copied ONCE, straight into the block buffer
  the device reads from (the library's page-locked input buffer) behind the overlap carried over from the previous block;
  there is no intermediate ring.  (The reference copies every sample three times: ZeroMQ frame -> ring buffer -> popped
  block -> page-locked buffer, sigFIFO.py:147-181 + demodulator_process.py:287.)
* ``RingBuffer`` -- the reference's standalone class (sigFIFO.py:13-103) for callers that use it directly: same contract
  (``insert`` appends a chunk and returns the fill level, ``popBlock(n)`` returns ``n`` samples or ``[]`` while fewer are
  buffered, ``flush``) and the same observable state (``headIdx``, ``tailIdx``, ``currentBufSize``; pinned by fixture G14),
  kept here as two running counters over a circular store.  One deliberate difference: a chunk that would overflow the
  buffer flushes it first and is then stored, which is what the reference's docstring and log message say (:44-54); the
  reference's code stores with the end index it computed before the flush and raises ValueError instead (recorded in G14,
  not reproduced).  A single chunk larger than the whole buffer cannot be stored and raises ValueError.
* ``SigFIFO.getBlock`` (sigFIFO.py:147-181) keeps feeding chunks into a ring until a block can be popped; chunks larger
  than the ring go in piece by piece.  The reference receives the chunks from a ZeroMQ SUB socket (GNU Radio sends ~4096
  samples at a time, the BER bench 2^14, examples/benchmark/bench_modem.py:32); transport is out of scope here, so the
  chunks come from any iterator and the end of the iterator plays the role of the cleared ``runStatus`` flag
  (``TimeoutError('Terminated')``, sigFIFO.py:167-169).
"""
import logging
import time

import numpy as np

log = logging.getLogger('pycusdr_amd.sigFIFO')


class BlockAssembler:
    """Blocks assembled in place.  ``buffer``: the N-sample block buffer (overlap included); ``push(chunk)`` yields the
    buffer every time it is complete -- the consumer must be done with it before it asks for the next one -- and carries the
    last ``overlap`` samples to the front afterwards."""

    def __init__(self, buffer, overlap):
        if not 0 <= overlap < len(buffer):
            raise IndexError('overlap must be shorter than the block')
        self.buf, self.ov = buffer, int(overlap)
        self.fill = self.ov              # the first block starts behind whatever the caller left in buffer[:overlap]
        self.blocks = 0

    def push(self, chunk):
        chunk = np.asarray(chunk)
        n, pos, size = len(chunk), 0, len(self.buf)
        while pos < n:
            take = min(n - pos, size - self.fill)
            self.buf[self.fill:self.fill + take] = chunk[pos:pos + take]
            self.fill += take
            pos += take
            if self.fill == size:
                self.blocks += 1
                self._moved = False
                yield self.buf
                if not self._moved:          # (a consumer that called retarget() has carried the overlap already)
                    self.buf[:self.ov] = self.buf[size - self.ov:]
                    self.fill = self.ov

    def retarget(self, other):
        """Continue in another buffer of the same length (called by the consumer while it holds a completed block): the
        overlap is carried over into ``other`` and the completed buffer is left untouched -- e.g. for a copy engine that is
        still reading it."""
        if len(other) != len(self.buf):
            raise IndexError('block buffers of unequal length')
        other[:self.ov] = self.buf[len(self.buf) - self.ov:]
        self.buf, self.fill, self._moved = other, self.ov, True


class WindowAssembler:
    """``BlockAssembler`` for a window of B consecutive blocks (the batched block path, mfb_receive_blocks_*): ``window`` holds
    ``B * stride + overlap`` samples, block b at ``b * stride`` (stride = block length - overlap: neighbours share their overlap
    as storage).  ``push(chunk)`` copies every sample once and yields B each time the window is complete; the consumer then calls
    ``retarget(other, nblocks)`` -- everything behind the ``nblocks`` blocks it took (the carried overlap, and whatever has been
    filled of the next block) moves to the front of ``other`` and filling goes on there, the completed window stays untouched for
    the copy engine.  ``complete_blocks()``: whole blocks in a partly filled window (end of a stream).

    Pull form: ``take(chunk)`` puts as much of the chunk as fits into the window and returns that count; ``full()``.  With a
    ``copier`` (``mfbank.HostCopy``) ``take`` only QUEUES the copy of a READ-ONLY chunk (``np.frombuffer`` of a received message
    is one; a replay marks its array ``flags.writeable = False``) for the library's copy thread -- the caller goes on (the host
    stages of the previous batch) and calls ``copier.drain()`` before it hands the window to the device.  A writable chunk may be
    storage its source fills again for the next chunk: it is copied on the spot (``copy_all_async``: queue those too).
    CONTRACT of the queued form: ``flags.writeable == False`` is taken as the source's promise that the samples stay as they
    are until the window has been handed to the device (``copier.drain()``).  A read-only VIEW over storage somebody else
    rewrites (a transport's reused receive ring) breaks that promise: such a source passes writable chunks, copies, or runs
    with ``"HIP": {"async_copies": false}``."""

    def __init__(self, window, overlap, stride, blocks, copier=None, copy_all_async=False):
        if len(window) != blocks * stride + overlap or overlap < 0 or stride < 1:
            raise IndexError('window of %d samples does not hold %d blocks of stride %d + overlap %d' % (len(window), blocks, stride, overlap))
        self.buf, self.ov, self.stride, self.B = window, int(overlap), int(stride), int(blocks)
        self.fill = self.ov              # the first block starts behind whatever the caller left in window[:overlap]
        self.stamps = []                 # time.time() at which each complete block of the window got its last sample
        self.copier = copier
        self.copy_all_async = bool(copy_all_async)

    def complete_blocks(self):
        return max(0, (self.fill - self.ov) // self.stride)

    def full(self):
        return self.fill == len(self.buf)

    def take(self, chunk):
        take = min(len(chunk), len(self.buf) - self.fill)
        if take:
            if self.copier is not None and (self.copy_all_async or not chunk.flags.writeable):
                self.copier.submit(self.buf, self.fill, chunk[:take])
            else:
                self.buf[self.fill:self.fill + take] = chunk[:take]
            self.fill += take
            done = (self.fill - self.ov) // self.stride
            if len(self.stamps) < done:
                now = time.time()
                self.stamps.extend([now] * (done - len(self.stamps)))
        return take

    def push(self, chunk):
        chunk = np.asarray(chunk)
        n, pos, size = len(chunk), 0, len(self.buf)
        while pos < n:
            take = min(n - pos, size - self.fill)
            self.buf[self.fill:self.fill + take] = chunk[pos:pos + take]
            self.fill += take
            pos += take
            while len(self.stamps) < (self.fill - self.ov) // self.stride:
                self.stamps.append(time.time())
            if self.fill == size:
                self._moved = False
                yield self.B
                if not self._moved:          # a consumer that is done with the window and keeps filling it
                    self.retarget(self.buf, self.B)

    def retarget(self, other, nblocks):
        if len(other) != len(self.buf):
            raise IndexError('windows of unequal length')
        start = int(nblocks) * self.stride
        rest = self.fill - start
        if nblocks < 0 or rest < self.ov:
            raise IndexError('the window does not hold %d complete blocks' % nblocks)
        if other is self.buf:
            other[:rest] = self.buf[start:self.fill].copy() if start < rest else self.buf[start:self.fill]
        else:
            other[:rest] = self.buf[start:self.fill]
        self.buf, self.fill, self._moved = other, rest, True
        del self.stamps[:int(nblocks)]


class RingBuffer:
    def __init__(self, outLen, bufLen=None, dtype=np.complex64):
        self.outLen = outLen
        if bufLen is None:
            bufLen = 10 * outLen
        elif bufLen < outLen:
            raise IndexError('bufLen < outLen', 'Buffer size too small for expected output size')
        self.bufLen = bufLen
        self.dtype = dtype
        self.buf = np.empty(self.bufLen, dtype=self.dtype)
        self._written = 0          # samples stored / consumed since the last flush
        self._read = 0

    # the reference's observable state, derived from the two counters
    @property
    def currentBufSize(self):
        return self._written - self._read

    @property
    def headIdx(self):            # one past the last stored sample, in [1, bufLen] once anything was stored
        return (self._written - 1) % self.bufLen + 1 if self._written else 0

    @property
    def tailIdx(self):
        return self._read % self.bufLen

    def _spans(self, counter, n):
        """The one or two contiguous pieces of the store that hold positions counter ... counter + n - 1."""
        start = counter % self.bufLen
        first = min(n, self.bufLen - start)
        return (start, first), (0, n - first)

    def insert(self, data):
        """Append a chunk; returns the number of buffered samples."""
        data = np.asarray(data)
        if data.dtype != self.dtype:
            log.error('chunk of %s handed to a %s buffer: converting', data.dtype, np.dtype(self.dtype))
            data = data.astype(self.dtype)
        n = len(data)
        if n > self.bufLen:
            raise ValueError(f'chunk of {n} samples cannot be stored in a buffer of {self.bufLen}')
        if self.currentBufSize + n > self.bufLen:
            log.error('chunk of %d samples does not fit beside the %d buffered ones: buffer flushed', n, self.currentBufSize)
            self.flush()
        (a, na), (b, nb) = self._spans(self._written, n)
        self.buf[a:a + na] = data[:na]
        if nb:
            self.buf[b:b + nb] = data[na:]
        self._written += n
        return self.currentBufSize

    def popBlock(self, noSamples):
        """``noSamples`` samples from the tail, or [] while fewer are buffered."""
        if self.currentBufSize < noSamples:
            return []
        (a, na), (b, nb) = self._spans(self._read, noSamples)
        if nb:
            out = np.concatenate((self.buf[a:a + na], self.buf[b:b + nb]))
        else:
            out = self.buf[a:a + na]        # a view, as in the reference: consume it before the next insert
        self._read += noSamples
        return out

    def flush(self):
        self._written = 0
        self._read = 0


class SigFIFO:
    """``getBlock()`` -> exactly ``reqDataSize`` samples, assembled from the chunks ``source`` yields."""

    def __init__(self, source, reqDataSize, dtype=np.complex64):
        self.blockSize = reqDataSize
        self.dtype = dtype
        self.source = iter(source)
        self.buf = RingBuffer(self.blockSize, bufLen=self.blockSize * 2, dtype=dtype)     # sigFIFO.py:141
        self._rest = None          # the part of an oversized chunk that has not gone in yet

    def getBlock(self):
        data = self.buf.popBlock(self.blockSize)
        while len(data) == 0:
            if self._rest is None:
                try:
                    self._rest = np.asarray(next(self.source))
                except StopIteration:
                    raise TimeoutError('Terminated') from None
            room = self.buf.bufLen - self.buf.currentBufSize
            if len(self._rest) <= room or (room == 0 and len(self._rest) <= self.buf.bufLen):
                # fits -- or fits after the flush the ring does when it is full, as the reference intends
                piece, self._rest = self._rest, None
            else:
                piece, self._rest = self._rest[:room], self._rest[room:]
            self.buf.insert(piece)
            data = self.buf.popBlock(self.blockSize)
        return data
