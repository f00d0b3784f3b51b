"""The per-block streaming loop around the hot path -- the caller of the Demodulator.

Mirrors the body of the reference's ``Demodulator_process.run`` (reference
demodulator_process.py:242-338): overlap carry in the page-locked input buffer, ``uploadAndFindCarrier``
then ``demodulate``, the result dict that goes to the decoder (same keys, DP:259-276), moving-average
timing and the rate line in ksamples/s (DP:186-190, 324-333).  The reference moves blocks over ZeroMQ
between OS processes; transport is out of scope here, so blocks come from any iterable and results
are handed to a callback (or returned) instead of ``zmq.send_pyobj``.
"""
import logging
import time

import numpy as np

from . import demodulator as demod_backends

log = logging.getLogger('pycusdr_amd.demodulator_process')


def radioBackendVoteGroupIDX(radioBackend):
    """Back-end name -> (module, vote group), as the reference maps it (DP:20-36): UHF 0, STX 1, STX1 2, STX2 3."""
    groups = {'UHF': (demod_backends.UHF, 0), 'STX': (demod_backends.STX, 1), 'STX1': (demod_backends.STX, 2),
              'STX2': (demod_backends.STX, 3), 'SBAND': (demod_backends.STX, 1)}
    if radioBackend not in groups:
        raise Exception('radioBackend {} not defined in voteGroup'.format(radioBackend))
    return groups[radioBackend]


class _NoCopier:
    """Stands in for mfbank.HostCopy when the chunk copies are made on the spot."""
    @staticmethod
    def drain():
        pass


class MarkedSource:
    """Chunks of a live transport with ``None`` where it would block (``DemodulatorRunner.drain_marked``)."""

    def __init__(self, poll, wait):
        self.poll, self.wait = poll, wait

    def __iter__(self):
        while True:
            chunk = self.poll()
            if chunk is None:
                yield None
                chunk = self.wait()
                if chunk is None:
                    return
            yield chunk


class DemodulatorRunner:
    """One receive channel, in-process.  ``feed(new_samples)`` takes exactly
    ``blockSize - overlap`` new complex64 samples and returns the result dict of that block."""

    def __init__(self, conf, protocol, radio, shard=None):
        self.conf, self.protocol, self.radioName = conf, protocol, radio
        self.confRadio = confRadio = conf['Radios']['Rx'][radio]
        confGPU = conf['GPU'][confRadio['CUDA_settings']]
        self.overlap = 2 ** confGPU['overlap']
        self.blockSize = 2 ** confGPU['blockSize']
        self.samplesPerSlice = self.blockSize - self.overlap
        self.baudRate = confRadio['baud']
        self.spSym = confRadio['samplesPerSym']
        self.Fs = self.baudRate * self.spSym
        worker_radio_name = confRadio.get('name', radio)
        self.workerId = conf['Main']['workerId'] + '-' + worker_radio_name
        self.radioBackend = confRadio['radioBackend']
        if 'voteGroup' in confRadio:
            self.demodulator = radioBackendVoteGroupIDX(self.radioBackend)[0]
            self.voteGroup = radioBackendVoteGroupIDX(confRadio['voteGroup'])[1]
        else:
            self.demodulator, self.voteGroup = radioBackendVoteGroupIDX(self.radioBackend)
        self.decoderProtocol = confRadio.get('Protocol', 'None')
        self.frequencyOffset_Hz = confRadio['frequencyOffset_Hz']
        self._fc = float(int(confRadio['frequency_Hz'] - confRadio['frequencyOffset_Hz']))    # the carrier the receiver is tuned to (DP:146)
        self.timeMA = 0.0
        self.iterCount = 0
        self.count = 0
        self.demod = self.demodulator.Demodulator(conf, protocol, radio, shard=shard)
        self.raw = self.demod.get_signalBufferHostPointer()
        self.raw[:] = 0

    def close(self):
        hc = getattr(self, '_copier', None)
        if hc is not None:
            hc.close()
            self._copier = None
        self.demod.close()

    def computeMATime(self, t):
        self.iterCount += 1
        self.timeMA = self.timeMA + (t - self.timeMA) / self.iterCount
        return self.timeMA

    def feed(self, new_samples):
        return self.feed_host(self.feed_device(new_samples))

    def feed_resident(self, device_ptr):
        """``feed_device`` for a whole block (N complex64 samples, overlap included) that already sits in device memory
        at ``device_ptr``: no host copy, no overlap carry."""
        if self.radioBackend != 'UHF':
            # the STX back end starts with peak clipping of the samples on the host (reference STX.py:13, DB:670-707)
            raise ValueError('device-resident blocks are only supported by the UHF back end: STX clips the samples on the host first')
        stamp = time.time()
        part = {'count': self.count, 'timestamp': stamp}
        part['doppler'], part['doppler_std'], _, part['SNR'] = self.demod.uploadAndFindCarrier(None, device_ptr=device_ptr)
        part['rec'] = self.demod.demodulateDevice()
        part['time_device'] = time.time() - stamp
        self.count += 1
        return part

    def feed_resident_begin(self, device_ptr):
        """``feed_resident`` in two halves (``feed_device_end`` collects): the block's device work is enqueued and the call returns."""
        if not (self.radioBackend == 'UHF' and getattr(self.demod, '_one_call', False)):
            self._flight = ('done', self.feed_resident(device_ptr))
            return
        stamp = time.time()
        self.demod.beginBlock(0, source='device', device_ptr=device_ptr)
        self._flight = ('flying', 0, self.count, stamp)
        self.count += 1

    def feed_device(self, new_samples=None):
        """Device half of one block (A3..A11): overlap carry, Doppler search, matched filters at the found shift, symbol
        decisions.  Carries no block-to-block state besides the overlap samples, so with time-chunk sharding
        (dist.BlockShard) any rank may run it for any block; the returned dict travels to the rank that runs
        ``feed_host`` in block order.  ``new_samples`` None: the block has been assembled in ``self.raw`` already and
        the assembler carries the overlap (run_stream)."""
        raw = self.raw
        in_place = new_samples is None
        if not in_place:
            if len(new_samples) != self.samplesPerSlice:
                raise ValueError(f'expected {self.samplesPerSlice} new samples per block, got {len(new_samples)}')
            raw[self.overlap:] = new_samples
        stamp = time.time()
        part = {'count': self.count, 'timestamp': stamp}
        part['doppler'], part['doppler_std'], _, part['SNR'] = self.demod.uploadAndFindCarrier(raw)
        if self.radioBackend == 'UHF':
            rec = self.demod.demodulateDevice()
        else:                                   # STX: fixed shift, no search (reference STX.py:21-24)
            self.demod.dopplerIdxlast = self.demod.doppOffsetIdx
            rec = self.demod.demodulateDevice()
        part['rec'] = rec
        part['time_device'] = time.time() - stamp
        if not in_place:
            raw[:self.overlap] = raw[-self.overlap:]      # overlap carry for the next block
        self.count += 1
        return part

    def _block_buffers(self):
        """The library's two page-locked input buffers, by the NAME the library knows them under ('pinned' = index 0,
        'pinned2' = index 1), and the index of the one ``self.raw`` -- where the next block is being assembled, carried
        overlap included -- currently is.  Buffer identity comes from the library, never from the order of earlier calls:
        a stream may end on either buffer and the next call must go on there."""
        bank = self.demod.bank
        bufs = (bank.input, bank.input2)
        if self.raw is not bufs[0] and self.raw.ctypes.data == bufs[1].ctypes.data:
            return bufs, 1
        if self.raw is not bufs[0] and self.raw.ctypes.data != bufs[0].ctypes.data:
            raise RuntimeError('the block buffer of this runner is not one of the library\'s page-locked input buffers')
        return bufs, 0

    def feed_device_begin(self, new_samples):
        """``feed_device`` in two halves: enqueue the block's device work and return at once (``feed_device_end`` collects it).
        One block in flight per runner.  Falls back to the synchronous call where the one-call block path is not in use (STX,
        Doppler-sharded handles, ``"one_call": false``)."""
        if len(new_samples) != self.samplesPerSlice:
            raise ValueError(f'expected {self.samplesPerSlice} new samples per block, got {len(new_samples)}')
        if not (self.radioBackend == 'UHF' and getattr(self.demod, '_one_call', False)):
            self._flight = ('done', self.feed_device(new_samples))
            return
        bufs, cur = self._block_buffers()
        raw = bufs[cur]
        raw[self.overlap:] = new_samples
        stamp = time.time()
        self.demod.beginBlock(cur, source=('pinned', 'pinned2')[cur])
        other = bufs[1 - cur]
        other[:self.overlap] = raw[-self.overlap:]      # overlap carry: the next block is assembled in the other buffer
        self.raw = other
        self._flight = ('flying', cur, self.count, stamp)
        self.count += 1

    def feed_device_end(self):
        fl, self._flight = self._flight, None
        if fl[0] == 'done':
            return fl[1]
        _, slot, count, stamp = fl
        part = {'count': count, 'timestamp': stamp}
        part['doppler'], part['doppler_std'], _, part['SNR'] = self.demod.endBlock(slot)
        part['rec'] = self.demod.demodulateDevice()
        part['time_device'] = time.time() - stamp
        return part

    def skip_block(self, new_samples):
        """A block another rank processes: keep the overlap carry and the block counter in step."""
        if len(new_samples) != self.samplesPerSlice:
            raise ValueError(f'expected {self.samplesPerSlice} new samples per block, got {len(new_samples)}')
        ov, sps = self.overlap, self.samplesPerSlice
        if ov <= sps:
            self.raw[:ov] = new_samples[-ov:]
        else:           # the overlap is longer than a slice: the tail of the old overlap stays part of the new one
            self.raw[:ov - sps] = self.raw[sps:ov].copy()
            self.raw[ov - sps:ov] = new_samples
        self.count += 1

    def feed_host(self, part, prev_tail=None, timed=True):
        """Sequential half of one block (A12, A13): bits, alignment against the previous block, trust tagging, and the
        result dict that goes to the decoder.  Must be called in block order -- or with the previous block's
        ``Demodulator.overlapTail`` as ``prev_tail`` (time-chunk sharding: the owner of a block runs this stage itself)."""
        t0 = time.time()
        bits, centres, trust, spSym = self.demod.demodulateHost(part['rec'], prev_tail=prev_tail)
        # (timed = False: the caller measured the whole loop turn itself -- the overlapped loops)
        spent = part['time_device'] + ((time.time() - t0) if timed else 0.0)
        return self.compose_result(part['count'], part['timestamp'], part['doppler'], part['doppler_std'], part['SNR'], bits, trust,
                                   spSym, spent)

    def compose_result(self, count, timestamp, doppler, doppler_std, SNR, bits, trust, spSym, spent):
        """The result dict of a block from its finished pieces: every key of the reference's dict (DP:259-276), including the
        two it initialises and never updates ('rangerateEst', 'baudRate_est': its loop writes 'rangerate' and 'baudrate_est'
        instead, DP:299,303)."""
        data = {'workerId': self.workerId, 'count': count, 'timestamp': timestamp, 'voteGroup': self.voteGroup,
                'doppler': doppler, 'doppler_std': doppler_std, 'data': bits, 'trust': trust, 'spSymEst': spSym, 'SNR': SNR,
                'rangerateEst': 0, 'baudRate': self.baudRate, 'baudRate_est': 0, 'sample_rate': self.Fs,
                'protocol': self.decoderProtocol}
        data['baudrate_est'] = self.Fs / data['spSymEst'] if data['spSymEst'] else 0.0
        # range rate implied by the measured frequency offset (reference computeTxFreqOffset, DP:359-379): against the carrier the
        # receiver is really tuned to, frequency_Hz minus the IF offset, as an integer (DP:146) -- pinned by fixture G19
        fc = self._fc
        data['rangerate'] = -data['doppler'] / fc * 299792458.0
        self.computeMATime(spent)
        data['time_ms'] = spent * 1e3
        data['rate_ksps'] = self.samplesPerSlice / spent / 1000
        data['rate_ksps_avg'] = self.samplesPerSlice / self.timeMA / 1000
        return data

    def report(self, d):
        """The reference's rate line (DP:324-333), every 50th block."""
        if d['count'] % 50 == 0:
            log.info('[%s]: freq offset % 6.0f Hz, SNR % 2.1f dB, est spsym % 3.2f, time % 3.2f ms (avg % 3.2f ms), '
                     'rate %5.0f ksamples/s (avg %5.0f)', self.radioName, d['doppler'], d['SNR'], d['spSymEst'],
                     d['time_ms'], self.timeMA * 1e3, d['rate_ksps'], d['rate_ksps_avg'])

    def blocks_per_call(self):
        """``"HIP": {"blocks_per_call": B}`` of the radio's GPU settings (next to the reference's "CUDA" block): B > 1 makes
        ``run_stream`` hand the device B consecutive blocks per call (mfb_receive_blocks_*), 1 is the reference's loop (one
        block per turn, DP:284-338).  Absent or "auto" (None here): ``run_stream`` batches what the source has ready -- up to
        ``auto_blocks_per_call()`` blocks while chunks come faster than the device takes them, every complete block at once
        while they do not."""
        confGPU = self.conf['GPU'][self.confRadio['CUDA_settings']]
        v = confGPU.get('HIP', {}).get('blocks_per_call', 'auto')
        return None if v in (None, 'auto') else max(1, int(v))

    def run_stream(self, chunk_source, sink=None, decoder=None, pipelined=False, overlapped=True, blocks_per_call=None):
        """The reference's loop shape (DP:284-338): chunks of ANY size (GNU Radio ~4096 samples, the BER bench
        2^14); ends when the chunk source is exhausted.  Sequential form: every chunk is copied once, straight into the
        page-locked input buffer behind the carried overlap (sigFIFO.BlockAssembler), and each completed block is processed
        in place; by default the device side of block i also overlaps the host stages of block i-1 (``overlapped``).
        Pipelined form: the stages run in threads, so blocks travel as copies through a SigFIFO."""
        if pipelined:
            from .sigFIFO import SigFIFO
            fifo = SigFIFO((c for c in chunk_source if c is not None), self.samplesPerSlice)

            def blocks():
                while True:
                    try:
                        yield fifo.getBlock()
                    except TimeoutError:
                        return
            return self.run(blocks(), sink=sink, decoder=decoder, pipelined=True)
        from .sigFIFO import BlockAssembler
        if not (overlapped and self.radioBackend == 'UHF' and getattr(self.demod, '_one_call', False)):
            asm = BlockAssembler(self.raw, self.overlap)
            return self.run((None for chunk in chunk_source if chunk is not None for _ in asm.push(chunk)), sink=sink, decoder=decoder)
        # how many blocks per device call: the caller's word, else the configuration's, else ("auto") whatever the source has ready.
        # A source that marks where it would block (MarkedSource) says so itself; a plain iterator cannot be asked, so the loop
        # watches how long each chunk takes to come (``_run_stream_batched``, adaptive): chunks that are there at once (a
        # recording, a backlog, a chunk of several blocks) fill windows of auto_blocks_per_call() blocks, and a block that
        # completes while the source is live (it made the loop wait less than a block of samples ago) goes out at once -- the
        # one-block loop's latency.
        B = max(1, int(blocks_per_call)) if blocks_per_call not in (None, 'auto') else self.blocks_per_call()
        adaptive = B is None and not isinstance(chunk_source, MarkedSource)
        if B is None:
            B = self.auto_blocks_per_call()
        if B > 1:       # (a batch of ONE block is slower than the one-block loop below: 211 against 254 Msamples/s at 2^15 x 64 -- more launches)
            bank = self.demod.bank
            if bank.get_search_path()['path'] == 'segment' and bank.get_search_mode() == 'transforms':
                return self._run_stream_batched(chunk_source, sink, decoder, B, adaptive=adaptive)
            # batches run on the segment search path (filters with a short impulse response: every shipped protocol) and the
            # default search mode: anything else takes the one-block loop
            if not adaptive:
                log.warning('[%s]: blocks_per_call = %d ignored: the filter bank / search mode of this handle runs one block per call', self.radioName, B)
        # Overlapped form: block i is on the device while this thread runs the sequential host stages and the decoder of
        # block i-1 and assembles block i+1 in the other page-locked buffer.  Same calls in the same order on the same data
        # as the plain loop, so the same results; only the waiting moves.
        self._apply_batch_overlap()
        bufs, cur = self._block_buffers()       # goes on in the buffer (and behind the overlap) the last call ended in
        names = ('pinned', 'pinned2')
        asm = BlockAssembler(bufs[cur], self.overlap)
        results, packets = [], []
        flying = None                    # (slot, count, timestamp)

        searching = []                   # the result dict whose bits the decoder is searching right now
        split = decoder is not None and hasattr(decoder, 'findFrames_begin')
        if split and hasattr(decoder, 'prepare'):
            decoder.prepare()

        def deliver(d):
            if sink is not None:
                sink(d)
            else:
                results.append(d)

        def finish_search():
            if searching:
                d = searching.pop()
                pk, _, nsync = decoder.findFrames_end()
                d['numSyncSig'] = nsync
                packets.extend(pk)
                deliver(d)

        last = [None]                    # when the previous block was collected

        def collect(fl):
            slot, count, stamp = fl
            part = {'count': count, 'timestamp': stamp}
            part['doppler'], part['doppler_std'], _, part['SNR'] = self.demod.endBlock(slot)
            part['rec'] = self.demod.demodulateDevice()
            now = time.time()
            # the loop's cost per block is the interval between two collects (the device side of block i overlaps the host stages
            # of block i - 1): that is what time_ms / rate_ksps report here (DP:324-333 times a loop turn); begin -> collect is
            # the block's latency
            part['time_device'] = (now - last[0]) if last[0] is not None else (now - stamp)
            last[0] = now
            d = self.feed_host(part, timed=False)
            d['latency_ms'] = (now - stamp) * 1e3
            self.report(d)
            if split:
                # the decoder's searches of this block run on the device while this thread goes on; their hits are
                # collected, and the packet state machine run, right before the next block's bits go in
                finish_search()
                decoder.findFrames_begin(d['data'], 0)
                searching.append(d)
                return
            if decoder is not None:
                pk, _, nsync = decoder.findFrames(d['data'], 0)
                d['numSyncSig'] = nsync
                packets.extend(pk)
            deliver(d)

        try:
            for chunk in chunk_source:
                if chunk is None:
                    # "nothing more right now" (see _run_stream_batched): the block in flight does not wait for the next one
                    if flying is not None:
                        fl, flying = flying, None
                        collect(fl)
                    finish_search()
                    continue
                for _ in asm.push(chunk):
                    self.demod.beginBlock(cur, source=names[cur])
                    started = (cur, self.count, time.time())
                    self.count += 1
                    cur = 1 - cur
                    # block i-1 lives in bufs[cur]: collect it (its host-to-device copy is then certainly over) BEFORE the
                    # overlap of block i is written into that buffer
                    if flying is not None:
                        fl, flying = flying, None
                        collect(fl)
                    asm.retarget(bufs[cur])
                    flying = started
            if flying is not None:
                fl, flying = flying, None
                collect(fl)
            finish_search()
        except BaseException:
            # leave nothing in flight behind a failure: the handle and the decoder must be usable for the next call
            for slot in (0, 1):
                try:
                    self.demod.bank.end_block(slot)
                except Exception:       # noqa: BLE001 -- nothing was in flight in this slot
                    pass
            if searching:
                try:
                    decoder.findFrames_end()
                except Exception:       # noqa: BLE001
                    pass
            raise
        finally:
            self.raw = bufs[cur]         # where the next block would be assembled
        return results, packets

    @staticmethod
    def drain_marked(poll, wait):
        """A chunk source for ``run_stream`` from a live transport: ``poll()`` returns the next chunk or None at once, ``wait()``
        blocks for the next chunk and returns None when the stream has ended.  Yields every chunk that is there, then a ``None``
        marker ("nothing more right now") before it blocks -- so a backlog goes through in batches of up to blocks_per_call
        blocks and a quiet source gets each block out as soon as it is complete.  With such a source batches cost no latency, so
        ``run_stream`` takes them without being asked (``auto_blocks_per_call``) unless the configuration says otherwise."""
        return MarkedSource(poll, wait)

    def _apply_batch_overlap(self):
        """``"HIP": {"batch_overlap": true}`` -> mfb_set_batch_overlap (a block / batch as two parts on two streams); off by default."""
        want = bool(self.conf['GPU'][self.confRadio['CUDA_settings']].get('HIP', {}).get('batch_overlap', False))
        if getattr(self, '_batch_overlap', False) != want and hasattr(self.demod.bank, 'set_batch_overlap'):
            self.demod.bank.set_batch_overlap(want)
            self._batch_overlap = want

    def auto_blocks_per_call(self):
        """Blocks per device call for a source that marks where it would block: windows of about 2^20 samples, at most 32 blocks
        (2^15-sample blocks: 32, 2^17: 8, 2^20 and above: one block per call) -- the sizes the sweep in profiles/r05_chain.md
        found best."""
        return max(1, min(32, (1 << 20) // self.blockSize))

    # a chunk that took longer than this to come had to be WAITED for: the source is live and has no backlog (a generator over a
    # recording answers in ~1 us, a queue with a backlog in a few; 4096 samples at 100 Msamples/s take 41 us to exist).  A replay
    # that spends longer than this on every chunk itself (reading, converting) looks live too and gets its blocks one by one --
    # correct, but slower than it could be: such a source says "blocks_per_call": B.
    PULL_SLOW_S = 25e-6

    def _run_stream_batched(self, chunk_source, sink, decoder, B, adaptive=False):
        """``run_stream`` with B consecutive blocks per device call (``"HIP": {"blocks_per_call": B}``; mfb_receive_blocks_*).
        The reference's loop hands the device one block per turn (DP:284-338); at its own block sizes (2^15 ... 2^17 samples,
        64 bins: config/base.json:13,33) a block is a few tens of microseconds of device work and the turn is all host.  Here the
        chunks are copied ONCE into a page-locked window of B blocks (block b at b * (N - overlap): the overlap between
        neighbours is shared storage, the carry happens once per window), the window goes through one set of launches while the
        host stages and the decoder of the previous window's blocks run, and the blocks come out one by one in stream order:
        the same result dicts, bits and packets as the one-block loop.  At the end of the source the complete blocks of the
        last, partly filled window are processed as a shorter batch -- and so they are whenever the source yields ``None``
        ("nothing more right now"): a live source that puts a ``None`` where it would block gets B blocks per call while it has
        a backlog and every block out as soon as it is complete while it has not (``drain_marked`` builds such a source from
        a poll function).  ``adaptive`` (a plain iterator, nothing configured): the loop decides the same thing from the time
        each ``next()`` took -- a block that completes while the source is live (a chunk had to be waited for, > PULL_SLOW_S, less
        than one block of samples ago) goes out at once with whatever else is complete; chunks that were simply there keep
        filling the window."""
        from .sigFIFO import WindowAssembler
        wins = self.demod.blockWindows(B)
        # "HIP": {"batch_overlap": true}: the next batch's search beside this batch's small kernels (mfb_set_batch_overlap).  Off
        # by default HERE: this loop is bound by its own per-block work and waits for batch k - 1 right after it has begun batch
        # k, and a batch k - 1 whose tail shares the chip with batch k's search arrives later (profiles/r06_chain.md: -7 %).
        self._apply_batch_overlap()
        names = ('window', 'window2')
        cur = 0
        # the integer stages behind the symbol decisions (bit lookup, block-overlap alignment, the decoder's searches on the
        # stream without a stash) on the device too, unless "HIP": {"stream_stages": false}: same results, bit for bit; blocks
        # the device flags as irregular go through the host code, and the device's state is then seeded again
        confGPU = self.conf['GPU'][self.confRadio['CUDA_settings']]
        stages = bool(confGPU.get('HIP', {}).get('stream_stages', True)) and hasattr(self.demod, 'enableStreamStages')
        if stages and B > 64:
            # the library runs the stream stages for batches of up to 64 blocks (include/mfbank.h): with more, every block would
            # come back without them, the chain would be marked dirty and re-seeded once per batch -- the host code does all of it
            log.info('[%s]: %d blocks per call: the integer stages stay on the host (the device takes them for batches of <= 64 blocks)',
                     self.radioName, B)
            stages = False
        if stages:
            key = id(decoder) if decoder is not None else None
            if getattr(self, '_stages_for', ()) != (key,):
                stages = self.demod.enableStreamStages(decoder if hasattr(decoder, 'findFrames_batch') else None)
                self._stages_for = (key,) if stages else ()
            if stages:
                stages = self.demod.seedStreamStages()
        wins[cur][:self.overlap] = self.raw[:self.overlap]      # goes on behind the overlap the last call left
        # the chunk -> window copies run on the library's copy thread (mfb_hostcopy_*) while this thread does the host stages of
        # the previous batch -- for chunks that cannot change under it: read-only arrays (np.frombuffer of a received message; a
        # replay marks its samples flags.writeable = False).  A writable chunk may be storage the source fills again: it is copied
        # on the spot.  "HIP": {"async_copies": true} queues every chunk (the source then keeps a window's worth of chunks
        # untouched), false none.
        mode = confGPU.get('HIP', {}).get('async_copies', 'auto')
        copier = _NoCopier
        if mode is not False:
            if getattr(self, '_copier', None) is None:
                from .mfbank import HostCopy
                self._copier = HostCopy()
            copier = self._copier
        asm = WindowAssembler(wins[cur], self.overlap, self.samplesPerSlice, B, copier=None if copier is _NoCopier else copier,
                              copy_all_async=mode is True)
        results, packets = [], []
        searching = []
        split = decoder is not None and hasattr(decoder, 'findFrames_begin')
        if split and hasattr(decoder, 'prepare'):
            decoder.prepare()
        last = [None]

        def deliver(d):
            if sink is not None:
                sink(d)
            else:
                results.append(d)

        def finish_search():
            if searching:
                d = searching.pop()
                pk, _, nsync = decoder.findFrames_end()
                d['numSyncSig'] = nsync
                packets.extend(pk)
                deliver(d)

        batch_dec = decoder is not None and hasattr(decoder, 'findFrames_batch')

        def host_stages(fl, record):
            """The host side of a finished batch: per-block estimates, A12 / A13 (the device's, or the host code for irregular
            blocks), the result dicts, the decoder."""
            slot, count0, nb, stamp, arrived = fl
            recs = self.demod.endBlocks(slot, record)
            now = time.time()
            per_block = ((now - last[0]) if last[0] is not None else (now - stamp)) / nb
            last[0] = now
            ds, ahead, edges = [], [], []
            for i, ((doppler, doppler_std, _, snr), rec) in enumerate(recs):
                part = {'count': count0 + i, 'timestamp': arrived[i], 'doppler': doppler, 'doppler_std': doppler_std, 'SNR': snr,
                        'rec': rec, 'time_device': per_block}
                d = self.feed_host(part, timed=False)
                d['latency_ms'] = (now - arrived[i]) * 1e3
                self.report(d)
                ds.append(d)
                # the hits of the decoder's searches came with the block -- valid while every block since the last seed took
                # the device's bits
                ahead.append(rec.get('_sync') if stages and not self.demod._stream_dirty else None)
                edges.append(rec.get('_edges') if ahead[-1] is not None else None)
            if batch_dec:
                # the decoder's searches of all blocks of the batch: delivered with the blocks, or one device round trip for
                # those that were not (Decoder.findFrames_batch)
                for d, (pk, _, nsync) in zip(ds, decoder.findFrames_batch([d['data'] for d in ds], 0, ahead=ahead if stages else None,
                                                                         edges=edges if stages else None)):
                    d['numSyncSig'] = nsync
                    packets.extend(pk)
                    deliver(d)
                return
            for d in ds:
                if split:
                    finish_search()
                    decoder.findFrames_begin(d['data'], 0)
                    searching.append(d)
                    continue
                if decoder is not None:
                    pk, _, nsync = decoder.findFrames(d['data'], 0)
                    d['numSyncSig'] = nsync
                    packets.extend(pk)
                deliver(d)

        END = object()
        it = iter(chunk_source)
        rest = [None]                    # what is left of a chunk that straddles two windows
        fill_s = [0.0]                   # how long the last fill() took: a source that is slower than the device is not kept waiting

        since_wait = [1 << 30]           # chunks pulled since one had to be waited for (adaptive)

        def fill():
            """Chunks from the source into the window (copies queued, not waited for) until it is complete ('full'), the source
            says it has nothing more right now ('dry') or ends ('end')."""
            t0 = time.time()
            live = False
            try:
                while True:
                    if rest[0] is None:
                        tp = time.perf_counter()
                        chunk = next(it, END)
                        if adaptive:
                            # live = the source made the loop wait while THIS block was coming in (sources hand over bursts: a
                            # producer's wake-up, a transport's message of several chunks -- the chunk that completes a block is
                            # often not the one that was waited for)
                            since_wait[0] = 0 if time.perf_counter() - tp > self.PULL_SLOW_S else since_wait[0] + 1
                        if chunk is END:
                            return 'end'
                        if chunk is None:
                            return 'dry'
                        rest[0] = chunk if isinstance(chunk, np.ndarray) else np.asarray(chunk)
                        live = adaptive and since_wait[0] * len(rest[0]) < self.samplesPerSlice
                    n = asm.take(rest[0])
                    rest[0] = rest[0][n:] if n < len(rest[0]) else None
                    if asm.full():
                        return 'full'
                    if live and asm.complete_blocks():
                        return 'dry'     # the source had nothing ready: what is complete goes out now
            finally:
                fill_s[0] = time.time() - t0

        flying = None                    # the batch on the device: (slot, first count, blocks, begin time, arrival stamps)
        waiting = None                   # a batch whose records are here and whose host stages have not run yet: (flight, record)
        try:
            state = fill()
            while True:
                # (host stages of the batch before last: beside the copies fill() has just queued)
                if waiting is not None:
                    w, waiting = waiting, None
                    host_stages(*w)
                copier.drain()
                nb = B if state == 'full' else asm.complete_blocks()
                if nb:
                    if stages and self.demod._stream_dirty:
                        # a block went through the host code: everything in flight was enqueued behind the stale state -- collect it,
                        # hand the device this side's state, go on
                        if flying is not None:
                            fl, flying = flying, None
                            host_stages(fl, self.demod.waitBlocks(fl[0]))
                        self.demod.seedStreamStages()
                    self.demod.beginBlocks(cur, nb, source=names[cur])
                    started = (cur, self.count, nb, time.time(), list(asm.stamps[:nb]))
                    self.count += nb
                    cur = 1 - cur
                    # the previous batch lives in wins[cur]: take its records (the device is then done with that window) BEFORE the
                    # samples behind this batch are carried into it
                    if flying is not None:
                        waiting = (flying, self.demod.waitBlocks(flying[0]))
                    asm.retarget(wins[cur], nb)
                    flying = started
                if state == 'full':
                    if fill_s[0] > 5e-3 and waiting is not None:
                        # a source slower than the device (a live radio): its results first, then wait for more samples
                        w, waiting = waiting, None
                        host_stages(*w)
                    state = fill()       # the next window's copies are on their way while the loop turns to `waiting`
                    continue
                # nothing more right now, or the end of the source: everything that is complete goes out
                if waiting is not None:
                    w, waiting = waiting, None
                    host_stages(*w)
                if flying is not None:
                    fl, flying = flying, None
                    host_stages(fl, self.demod.waitBlocks(fl[0]))
                if not batch_dec:
                    finish_search()
                if state == 'end':
                    break
                state = fill()
        except BaseException:
            for slot in (0, 1):
                try:
                    self.demod.bank.end_blocks(slot)
                except Exception:       # noqa: BLE001 -- nothing was in flight in this slot
                    pass
            if searching:
                try:
                    decoder.findFrames_end()
                except Exception:       # noqa: BLE001
                    pass
            raise
        finally:
            # the overlap a later call (batched or not) goes on behind; samples of an incomplete block are dropped, as the
            # one-block loop drops them
            try:
                copier.drain()
            finally:
                self.raw[:self.overlap] = asm.buf[:self.overlap]
        return results, packets

    def run(self, sample_source, sink=None, decoder=None, pipelined=False):
        """Drive the loop over an iterable of new-sample slices.  With a ``decoder`` every block's
        bits go through ``findFrames`` and the packets are collected.

        ``pipelined``: the three stages the reference runs as three OS processes joined by ZeroMQ (sample source ->
        Demodulator_process -> decoder process, pyCuSDR.py / DP:242-338) run as three threads joined by bounded queues:
        the ring buffer assembles block i+1 and the decoder works on block i-1 while the device searches block i
        (the library calls release the GIL).  Same results in the same order as the sequential loop."""
        results, packets = [], []

        def decode(d):
            if decoder is not None:
                pk, _, nsync = decoder.findFrames(d['data'], 0)
                d['numSyncSig'] = nsync
                packets.extend(pk)
            if sink is not None:
                sink(d)
            else:
                results.append(d)

        report = self.report

        if not pipelined:
            for chunk in sample_source:       # None: the block sits in self.raw already (run_stream)
                d = self.feed(None if chunk is None else np.asarray(chunk, dtype=np.complex64))
                decode(d)
                report(d)
            return results, packets

        import queue
        import sys
        import threading
        END = object()
        # the stages hand the interpreter lock to each other once per block: with the default 5 ms switch interval a
        # stage that returns from a library call can wait that long for a stage busy in Python code
        old_switch = sys.getswitchinterval()
        sys.setswitchinterval(min(old_switch, 1e-4))
        q_in, q_out = queue.Queue(maxsize=2), queue.Queue(maxsize=4)
        failure = []

        def stage(body, q_to):
            """Run ``body``; whatever happens, tell the next stage that the stream has ended."""
            def target():
                try:
                    body()
                except BaseException as e:      # noqa: BLE001 -- re-raised in the caller's thread below
                    failure.append(e)
                finally:
                    if q_to is not None:
                        q_to.put(END)
            t = threading.Thread(target=target, daemon=True)
            t.start()
            return t

        def source_body():
            for chunk in sample_source:
                if failure:
                    return
                q_in.put(np.array(chunk, dtype=np.complex64))      # own copy: the ring buffer reuses its storage

        def decoder_body():
            while True:
                d = q_out.get()
                if d is END:
                    return
                if not failure:                 # after a failure keep draining so that the producer never blocks
                    try:
                        decode(d)
                    except BaseException as e:  # noqa: BLE001
                        failure.append(e)

        t_src = stage(source_body, q_in)
        t_dec = stage(decoder_body, None)
        try:
            while True:
                chunk = q_in.get()
                if chunk is END:
                    break
                if failure:
                    continue                                        # drain so that the source thread can finish
                d = self.feed(chunk)
                q_out.put(d)
                report(d)
        except BaseException as e:                                   # noqa: BLE001
            failure.append(e)
            while q_in.get() is not END:                            # let the source thread run out
                pass
        finally:
            q_out.put(END)
            t_dec.join()
            t_src.join()
            sys.setswitchinterval(old_switch)
        if failure:
            raise failure[0]
        return results, packets
