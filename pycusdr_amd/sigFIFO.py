"""Fixed-size blocks out of arbitrary-size chunks: the buffer half of the reference's sigFIFO.py.

``RingBuffer`` keeps the reference's semantics (sigFIFO.py:13-103, pinned by fixture G14): ``insert`` appends
a chunk, ``popBlock(n)`` returns ``n`` samples or an empty list while fewer are buffered (:70-75).  One
deliberate difference: a chunk that would overflow the buffer flushes it first and is then stored, which is
what the reference's docstring and log message say (:44-54); the reference's code stores with the end index
it computed before the flush and raises ValueError instead (recorded in G14, not reproduced).  ``SigFIFO.getBlock`` (sigFIFO.py:147-181) keeps
feeding chunks into the ring until a block can be popped.  The reference receives the chunks from a ZeroMQ SUB
socket (GNU Radio sends ~4096 samples at a time, the BER bench 2^14, examples/benchmark/bench_modem.py:32);
transport is out of scope here, so the chunks come from any iterator and the end of the iterator plays the role
of the cleared ``runStatus`` flag (``TimeoutError('Terminated')``, sigFIFO.py:167-169).
"""
import logging

import numpy as np

log = logging.getLogger('pycusdr_amd.sigFIFO')


class RingBuffer:
    def __init__(self, outLen, bufLen=None, dtype=np.complex64):
        self.outLen = outLen
        if bufLen is None:
            self.bufLen = 10 * outLen
        else:
            if bufLen < outLen:
                raise IndexError('bufLen < outLen', 'Buffer size too small for expected output size')
            self.bufLen = bufLen
        self.dtype = dtype
        self.buf = np.empty(self.bufLen, dtype=self.dtype)
        self.headIdx = 0
        self.tailIdx = 0
        self.currentBufSize = 0

    def insert(self, data):
        """Append a chunk; returns the number of buffered samples."""
        data = np.asarray(data)
        if data.dtype != self.dtype:
            log.error('wrong datatype. Expected %s', self.dtype)
            data = data.astype(self.dtype)
        n = len(data)
        if self.currentBufSize + n > self.bufLen:
            log.error('buffer full: Flush')
            self.flush()
        end = self.headIdx + n
        if end > self.bufLen:
            first = self.bufLen - self.headIdx
            self.buf[self.headIdx:] = data[:first]
            self.headIdx = n - first
            self.buf[:self.headIdx] = data[first:]
        else:
            self.buf[self.headIdx:end] = data
            self.headIdx = end
        self.currentBufSize += n
        return self.currentBufSize

    def popBlock(self, noSamples):
        """``noSamples`` samples from the tail, or [] while fewer are buffered."""
        if self.currentBufSize < noSamples:
            return []
        end = self.tailIdx + noSamples
        if end > self.bufLen:
            first = self.bufLen - self.tailIdx
            data = np.empty(noSamples, dtype=self.dtype)
            data[:first] = self.buf[-first:]
            self.tailIdx = noSamples - first
            data[first:] = self.buf[:self.tailIdx]
        else:
            data = self.buf[self.tailIdx:end]         # a view, as in the reference: consume it before the next insert
            self.tailIdx = 0 if end == self.bufLen else end
        self.currentBufSize -= noSamples
        return data

    def flush(self):
        self.headIdx = 0
        self.tailIdx = 0
        self.currentBufSize = 0


class SigFIFO:
    """``getBlock()`` -> exactly ``reqDataSize`` samples, assembled from the chunks ``source`` yields."""

    def __init__(self, source, reqDataSize, dtype=np.complex64):
        self.blockSize = reqDataSize
        self.dtype = dtype
        self.source = iter(source)
        self.buf = RingBuffer(self.blockSize, bufLen=self.blockSize * 2, dtype=dtype)     # sigFIFO.py:141

    def getBlock(self):
        data = self.buf.popBlock(self.blockSize)
        while len(data) == 0:
            try:
                chunk = next(self.source)
            except StopIteration:
                raise TimeoutError('Terminated') from None
            self.buf.insert(np.asarray(chunk))
            data = self.buf.popBlock(self.blockSize)
        return data
