"""Multi-GPU Doppler-bin sharding: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" on CPU for tests).

The reference has no distributed runtime.  The hot path shards by Doppler bin:
rank r of G owns the contiguous slice [r*D/G, (r+1)*D/G) of the bin table and holds a full replica of
the filter bank.  Per block:
  1. rank 0 owns the IQ stream; its N-sample block is broadcast to every rank (8 MiB over xGMI) --
     ranks need no feeder process of their own.  The broadcast of block i+1 can be started while block i is
     searched (``prefetch``: a communication stream and a communicator of its own, two block buffers), so in
     a stream of blocks the transfer costs nothing;
  2. every rank searches its bins (libmfbank, same stream);
  3. ONE collective on the per-bin scores: an all-gather of the ranks' slices when the bins divide evenly
     (the C4 layout), otherwise an all-reduce (sum) in which every rank has zeros outside its slice -- adding
     exact zeros; either way the table is bit-identical on all ranks.  With SUM_ALL_MASKS (every shipped
     protocol) only column 0 of doppSum is populated (reference cuda_kernels.cu:453-464), so only that
     column travels: D floats instead of D*M;
  4. the Doppler pick on the full table on every rank (identical everywhere);
  5. the demodulation stage (1/D of the work) runs on EVERY rank: each one holds the block, the full filter bank and
     the pick, so every rank returns the same bits and carries the same block-to-block state (checkSymbolOverlap's
     windows DB:977-979, the decoder's overlap buffer DEC:89-90) whichever slice the carrier drifts through.
The exchange is latency-bound (<= 8 KiB), not link-bandwidth-bound.
"""
import contextlib
import logging

import numpy as np

log = logging.getLogger('pycusdr_amd.dist')


def bin_slice(num_bins, rank, world):
    """Contiguous, near-equal partition of ``num_bins`` over ``world`` ranks."""
    base, rem = divmod(num_bins, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def bin_owner(num_bins, world, bin_index):
    """Rank whose slice holds ``bin_index``."""
    for r in range(world):
        lo, hi = bin_slice(num_bins, r, world)
        if lo <= bin_index < hi:
            return r
    raise ValueError(f'bin {bin_index} outside [0, {num_bins})')


class StepWatchdog:
    """Turns a hang of the sharded loop into a diagnosis and a NON-ZERO EXIT.  A collective that some rank never joins does
    not fail, it waits -- RCCL inside a kernel on the device (the host notices at the next read-back), gloo inside the call.
    The loop calls ``beat()`` once per step; a daemon thread that sees no beat for ``timeout_s`` seconds writes, to stderr,
    the rank, the step, what the shard was doing (``describe()``: the phase and the collectives enqueued since the last
    completed synchronisation) and leaves the process with ``exit_code`` -- the launcher then tears the other ranks down.
    Leaving is ``os._exit``: nothing is re-executed, so it is safe in a process that has touched the GPU."""

    def __init__(self, timeout_s, rank=0, describe=None, exit_code=3, stream=None, first_grace_s=300.0):
        """``first_grace_s``: until the first beat the limit is at least this long -- the first step of a job creates the
        communicators and loads the code objects."""
        import threading
        import time
        self.timeout_s, self.rank, self.describe, self.exit_code = float(timeout_s), int(rank), describe, int(exit_code)
        self.first_grace_s, self._beats = float(first_grace_s), 0
        self._time, self._last, self.step, self._stop = time, time.monotonic(), -1, threading.Event()
        self._out = stream
        self._thread = threading.Thread(target=self._watch, name='mfb-step-watchdog', daemon=True)
        if self.timeout_s > 0:
            self._thread.start()

    def beat(self, step=None):
        self._beats += 1
        self._suspended = False
        self._last = self._time.monotonic()
        self.step = self.step + 1 if step is None else int(step)

    def stop(self):
        self._stop.set()

    def suspend(self):
        """Stop counting: the loop that beats has returned (``BlockShard.run`` / ``GridShard.run`` suspend the watchdog they were
        given when they leave, so a caller that spends longer than ``timeout_s`` on its results, or idles before the next
        ``run``, is not killed with a misleading 'no progress').  ``resume()`` / the next ``beat()`` re-arm it.  The caller
        owns ``stop()``."""
        self._suspended = True

    def resume(self):
        self._last = self._time.monotonic()
        self._suspended = False

    def _watch(self):
        import os
        import sys
        while not self._stop.wait(min(0.25, self.timeout_s / 4)):
            if getattr(self, '_suspended', False):
                continue
            idle = self._time.monotonic() - self._last
            if idle > (self.timeout_s if self._beats else max(self.timeout_s, self.first_grace_s)):
                what = ''
                try:
                    what = self.describe() if self.describe is not None else ''
                except Exception as e:      # noqa: BLE001 -- the diagnosis must not keep the process alive
                    what = f'(no description: {e})'
                out = self._out or sys.stderr
                out.write(f'[mfb watchdog] rank {self.rank}: no progress for {idle:.1f} s after step {self.step}; {what}\n')
                out.flush()
                os._exit(self.exit_code)


class DopplerShard:
    """Glue between a bank (MFBank, or any object with its device-pointer methods) holding this rank's
    bins and the process group.  Works on CUDA/HIP tensors over RCCL and on CPU tensors over gloo."""

    def __init__(self, rank=None, world=None, group=None, device=None, comm=None, src=0, bcast_group=None,
                 concurrent_broadcast=True):
        """``concurrent_broadcast`` False: SINGLE-COMMUNICATOR mode -- the block broadcast uses the communicator and the
        stream of the score exchange, so every rank issues every collective of the job in one program order on one stream
        (what NCCL / RCCL guarantee to be deadlock-free); a prefetched block then queues behind the running search instead
        of travelling beside it.  True (default): the broadcast has a communicator and a stream of its own.
        ``bcast_group``: a second communicator over the same ranks for the block broadcast, so that it runs beside the
        exchange of the scores instead of queueing behind it (made here when ``group`` is None).  ``src``: the PROCESS rank of this shard's rank 0 (the broadcast source; differs from 0 when ``group`` is a
        subset of the job, GridShard).  ``comm``: an object with torch.distributed's call surface (get_rank, get_world_size, get_backend, new_group,
        broadcast, all_gather_into_tensor, all_reduce, ReduceOp); default torch.distributed itself.  A caller that
        runs several ranks inside one process (rehearsals of many-rank geometries on one device) passes its own."""
        import torch
        if comm is None:
            import torch.distributed as comm
        self.torch, self.dist = torch, comm
        dist = comm
        self.group = group
        self.src = int(src)
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.backend = dist.get_backend(group)
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) if self.backend == 'nccl' else torch.device('cpu')
        self.device = torch.device(device)
        self.on_gpu = self.device.type == 'cuda'
        self.scores = None
        self.block = None
        # a dedicated side stream: the bank's kernels, the buffer clear and the collectives are all
        # ordered on it (an RCCL op is synchronised against the *current* torch stream)
        self.stream = torch.cuda.Stream(self.device) if self.on_gpu else None
        # block distribution ahead of time: its own stream and its own communicator, so that the broadcast of
        # the next block runs beside the search and the all-reduce of the current one
        self.concurrent_broadcast = bool(concurrent_broadcast)
        if not self.concurrent_broadcast:
            self.comm, self.bcast_group = self.stream, group
        else:
            self.comm = torch.cuda.Stream(self.device) if self.on_gpu else None
            if bcast_group is not None:
                self.bcast_group = bcast_group
            else:
                self.bcast_group = dist.new_group(backend=self.backend) if (self.world > 1 and group is None) else group
        # for the watchdog: what this rank is doing, and the collectives enqueued since the last completed synchronisation
        import collections
        self.phase, self.ops = 'idle', collections.deque(maxlen=8)

    def describe(self):
        return (f'phase: {self.phase}; collectives enqueued since the last completed synchronisation: '
                f'{list(self.ops) or "none"} (backend {self.backend}, world {self.world}, '
                f'{"own" if self.concurrent_broadcast else "shared"} broadcast communicator)')

    def _on_stream(self, which=None):
        which = self.stream if which is None else which
        return self.torch.cuda.stream(which) if self.on_gpu else contextlib.nullcontext()

    def bin_range(self, num_bins):
        return bin_slice(num_bins, self.rank, self.world)

    def owner(self, table_index):
        """Rank whose slice holds row ``table_index`` of the exchanged table (noise rows: rank 0)."""
        b = int(table_index) - getattr(self, 'doff', 0)
        return 0 if b < 0 else bin_owner(self.D, self.world, b)

    def attach(self, bank, num_bins_total, M, sum_all=False, noise_rows=0, exchange='auto'):
        """Allocate the exchange buffers and run the bank on the shard's stream so that the collectives
        are ordered after the search without host synchronisation.

        ``exchange``: 'allgather' / 'allreduce' force one form of the exchange (default: all-gather whenever the slices
        are equal).  ``noise_rows``: rows in front of the bin table that every rank computes itself -- the reference's
        noise-reference bin (DB:148-159; its score divides the metric, CU:550-554).  The table every rank picks on is
        [noise rows | all D bins]; the noise rows are taken from rank 0 so that it is bit-identical everywhere."""
        torch = self.torch
        self.D, self.M, self.sum_all, self.doff = int(num_bins_total), int(M), bool(sum_all), int(noise_rows)
        rows = self.doff + self.D
        shape = (rows,) if self.sum_all else (rows, self.M)
        self.scores = torch.zeros(shape, dtype=torch.float32, device=self.device)
        # equal slices: every rank exports into a slice-sized buffer and the table is all-gathered (no clearing,
        # no additions); uneven slices: all-reduce of the zero-padded table
        # (gloo moves device tensors through host staging and has no all_gather_into_tensor for them: all-reduce there)
        if exchange not in ('auto', 'allgather', 'allreduce'):
            raise ValueError(f'exchange must be auto, allgather or allreduce, got {exchange!r}')
        can_gather = self.D % self.world == 0 and (self.backend != 'gloo' or not self.on_gpu)
        if exchange == 'allgather' and not can_gather:
            raise ValueError('all-gather needs equal slices (and, over gloo, CPU tensors)')
        self.even = can_gather and exchange != 'allreduce'
        self.nloc = self.D // self.world if self.even else None
        lrows = self.doff + self.D // self.world
        self.local = torch.zeros((lrows,) + shape[1:], dtype=torch.float32, device=self.device) if self.even else None
        # with noise rows the gathered layout is [rank][noise rows | slice]; it is re-packed into the table afterwards
        self.gathered = (torch.zeros((self.world * lrows,) + shape[1:], dtype=torch.float32, device=self.device)
                         if self.even and self.doff else None)
        # complex64 blocks, interleaved: two buffers so that one can be filled while the other is searched
        self.blocks = [torch.empty(2 * bank.N, dtype=torch.float32, device=self.device) for _ in range(2)]
        self.block = self.blocks[0]
        self.cur, self.pending = 0, None
        self.ready = [None, None]      # events: buffer k holds a complete block
        self.free = [None, None]       # events: the search that read buffer k has been enqueued and finished
        if self.on_gpu:
            torch.cuda.synchronize(self.device)
            bank.set_stream(self.stream.cuda_stream)

    def full_scores(self):
        """The exchanged score table as float32 [noise rows + D, M] (column 0 only is populated under SUM_ALL_MASKS)."""
        s = self.scores.detach().cpu().numpy()
        if not self.sum_all:
            return s
        out = np.zeros((self.doff + self.D, self.M), dtype=np.float32)
        out[:, 0] = s
        return out

    def _event(self, slot, k, stream):
        """Record on ``stream`` the event of buffer k in ``slot`` (self.ready / self.free); the two events per slot are
        created once and re-armed."""
        if slot[k] is None:
            slot[k] = self.torch.cuda.Event()
        slot[k].record(stream)

    def _fill(self, k, block, stream):
        """Enqueue on ``stream``: buffer k <- rank 0's block, on every rank."""
        with self._on_stream(stream):
            if self.on_gpu and self.free[k] is not None:
                stream.wait_event(self.free[k])               # the previous search on this buffer is done
            if self.rank == 0:
                if block is None:
                    raise ValueError('rank 0 must supply the block')
                self.blocks[k].copy_(block.reshape(-1), non_blocking=True)
            if self.world > 1:
                self.phase = f'broadcast of a block into buffer {k} from rank {self.src}'
                self.ops.append(f'broadcast(buffer {k})')
                self.dist.broadcast(self.blocks[k], src=self.src, group=self.bcast_group)
            if self.on_gpu:
                self._event(self.ready, k, stream)

    def prefetch(self, block=None):
        """Start distributing the NEXT block (rank 0 passes it) while the current one is being searched."""
        k = 1 - self.cur
        self._fill(k, block, self.comm if self.on_gpu else None)
        self.pending = k

    def broadcast_block(self, bank, block=None):
        """Make rank 0's block (a float32 view of N complex64 samples on the shard's device) the bank's input on
        every rank and run the forward FFT on it.  Uses the block a preceding ``prefetch`` distributed, if any."""
        if self.pending is not None:
            k, self.pending = self.pending, None
        else:
            k = 1 - self.cur
            self._fill(k, block, self.stream)
        self.cur = k
        self.block = self.blocks[k]
        # the bank launches on the shard's stream by itself (attach: set_stream); torch's stream context is entered only
        # around torch's own operations -- every entry and exit costs the host several microseconds the device idles for
        if self.on_gpu and self.ready[k] is not None:
            self.stream.wait_event(self.ready[k])
        bank.upload_device(self.blocks[k].data_ptr())

    def search_and_pick(self, bank, row_offset, after_search=None):
        """Search this rank's bins, exchange, pick.  ``row_offset`` = first bin of this rank's slice (bin_range()[0]).
        ``after_search()`` runs on the host right after the search has been enqueued (the device is busy from then on):
        the place to start distributing the next block."""
        col = self.sum_all
        doff = self.doff
        if not self.even:
            with self._on_stream():
                self.scores.zero_()
        bank.search_async()
        if self.on_gpu:                           # from here on the block buffer may be refilled
            self._event(self.free, self.cur, self.stream)
        if after_search is not None:
            after_search()
        if self.even:
            # [noise rows | slice] of every rank, gathered; without noise rows that IS the table
            bank.export_rows_async(self.local.data_ptr(), 0, 0, doff + self.nloc, column_only=col)
            with self._on_stream():
                self.phase = 'all-gather of the per-bin scores'
                self.ops.append('all_gather(scores)')
                if doff:
                    self.dist.all_gather_into_tensor(self.gathered, self.local, group=self.group)
                    g = self.gathered.view((self.world, doff + self.nloc) + tuple(self.gathered.shape[1:]))
                    self.scores[:doff].copy_(g[0, :doff])                                   # rank 0's noise rows
                    self.scores[doff:].view((self.world, self.nloc) + tuple(self.scores.shape[1:])).copy_(g[:, doff:])
                else:
                    self.dist.all_gather_into_tensor(self.scores, self.local, group=self.group)
        else:
            nloc = bank.D
            bank.export_rows_async(self.scores.data_ptr(), doff + row_offset, doff, nloc, column_only=col)
            if doff and self.rank == 0:
                bank.export_rows_async(self.scores.data_ptr(), 0, 0, doff, column_only=col)
            with self._on_stream():
                self.phase = 'all-reduce of the per-bin scores'
                self.ops.append('all_reduce(scores)')
                self.dist.all_reduce(self.scores, op=self.dist.ReduceOp.SUM, group=self.group)     # adds exact zeros
        self.phase = 'pick read-back (waits on the stream for the block, the search and the exchange of the scores)'
        if self.sum_all:
            res = bank.pick_column(self.scores.data_ptr(), num=self.D, offset=doff)
        else:
            res = bank.pick(self.scores.data_ptr(), num=self.D, offset=doff)
        # the read-back synchronised the shard's stream: everything enqueued on it has completed (a prefetched broadcast on
        # the side stream, started after the search, may still be in flight -- it stays listed)
        pending = [o for o in self.ops if o.startswith('broadcast') and self.pending is not None and self.concurrent_broadcast][-1:]
        self.ops.clear()
        self.ops.extend(pending)
        self.phase = 'between steps'
        return res

    def step(self, bank, row_offset, block=None, next_block=None, prefetch_next=False):
        """One block of the sharded hot path: (take the prefetched block or broadcast now), start the next block's
        broadcast if the caller knows it, search, exchange, pick."""
        self.broadcast_block(bank, block)
        # the search goes out first; the next block's broadcast is started while it runs
        return self.search_and_pick(bank, row_offset, after_search=(lambda: self.prefetch(next_block)) if prefetch_next else None)


def allreduce_scores_host(local_scores, row_offset, num_bins_total, group=None):
    """CPU/gloo statement of the exchange on numpy arrays (documentation of the collective): returns the
    full [D_total, M] matrix on every rank."""
    import torch
    import torch.distributed as dist
    local_scores = np.asarray(local_scores, dtype=np.float32)
    full = torch.zeros((num_bins_total, local_scores.shape[1]), dtype=torch.float32)
    full[row_offset:row_offset + local_scores.shape[0]] = torch.from_numpy(local_scores)
    dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
    return full.numpy()


# ---- time-chunk sharding ---------------------------------------------------------------------------------------------
_TAIL_DTYPES = (np.uint8, np.bool_, np.float64)


class BlockShard:
    """Blocks dealt round-robin over the ranks (SURVEY 8e "Alternative"; best when one GPU holds the whole bin table):
    rank r runs ALL stages but the decoder of blocks r, r + G, ... -- the device stages (forward FFT, Doppler search, pick,
    matched filters at the found shift, symbol decisions: A3..A11) and the host stages (bit lookup, alignment against the
    previous block, trust tagging: A12 / A13).  The alignment of block i looks at block i - 1 only through its ``overlapTail``
    -- the ~50 bits around the end of its window (reference demodulator_base.py:977-979), a function of that block's device
    results alone -- so the owner of block i - 1 forwards those bits to the owner of block i (one small point-to-point message),
    and nobody waits for a chain.  The ROOT receives finished bit slices in block order and runs only what really is
    sequential: the decoder with its overlap buffer (reference decoder.py:89-90).  No collective on the data path.

    (Round 3 ran A12 / A13 on the root: 0.24 ms per C2 block of sequential work, a ceiling of about six GPUs' worth of blocks.
    What is left on the root is the decoder, 0.05 ms per block.)

    Every rank reads the same sample stream (in the reference any number of processes may subscribe to the SDR's ZeroMQ
    publisher, pyCuSDR.py:245-251 starts one Demodulator_process per radio on it); a rank copies only the overlap tail of
    the blocks it does not own.  ``comm``: torch.distributed's call surface (send, recv, isend, get_rank, get_world_size);
    ``group``: a process group whose backend moves CPU tensors (gloo) -- with an RCCL default group pass
    ``torch.distributed.new_group(backend='gloo')``.
    """
    HEADER = 12         # float64: block index, doppler, doppler_std, SNR, spSym, kept symbols, device + host seconds, tail: post, end,
    #                     dtype code, exact flag; the owner's block stamp (epoch seconds)
    TAIL_HEADER = 5     # float64: block index, post, end, dtype code, exact flag

    def __init__(self, rank=None, world=None, group=None, comm=None, root=0, ranks=None):
        """``ranks``: the process (global rank) behind each member of this shard's world, for worlds that are a subset of
        the job (GridShard: one member per bin group); default: member r is process r."""
        import torch
        if comm is None:
            import torch.distributed as comm
        self.torch, self.dist, self.group = torch, comm, group
        self.rank = comm.get_rank(group) if rank is None else int(rank)
        self.world = comm.get_world_size(group) if world is None else int(world)
        self.root = int(root)
        if not 0 <= self.root < self.world:
            raise ValueError(f'root {root} outside the world of {self.world} ranks')
        self.peers = list(range(self.world)) if ranks is None else [int(r) for r in ranks]
        if len(self.peers) != self.world:
            raise ValueError(f'{len(self.peers)} process ranks for a world of {self.world}')
        self._inflight = []
        self.phase = 'idle'

    def describe(self):
        return f'phase: {self.phase} (time-block shard: rank {self.rank} of {self.world}, root {self.root})'

    def owner(self, block_index):
        return int(block_index) % self.world

    # -- wire formats ----------------------------------------------------------------------------------------------------
    @staticmethod
    def pack_tail(index, tail):
        """Owner of block ``index`` -> owner of the next block: a fixed header, then the two bit runs as bytes."""
        post, end = np.asarray(tail['post']), np.asarray(tail['end'])
        code = [i for i, t in enumerate(_TAIL_DTYPES) if post.dtype == t]
        if not code or end.dtype != post.dtype:
            raise TypeError(f'bit arrays of type {post.dtype} / {end.dtype} cannot travel')
        head = np.array([index, len(post), len(end), code[0], 1.0 if tail['exact'] else 0.0], dtype=np.float64)
        return head, np.concatenate((post.astype(np.uint8), end.astype(np.uint8)))

    @staticmethod
    def unpack_tail(head, body):
        n_post, n_end, dt = int(head[1]), int(head[2]), _TAIL_DTYPES[int(head[3])]
        return int(head[0]), {'post': body[:n_post].astype(dt), 'end': body[n_post:n_post + n_end].astype(dt), 'exact': bool(head[4])}

    def pack(self, d, tail):
        """Owner -> root: the finished block -- estimates, the kept symbols' bits / centres / trust (uint8 each, as the
        caller gets them, DB:859), and the block's tail (the root carries it as its own alignment state)."""
        bits, trust = np.ascontiguousarray(d['data'], dtype=np.uint8), np.ascontiguousarray(d['trust'], dtype=np.uint8)
        if len(bits) != len(trust):
            raise ValueError('bit and trust arrays of unequal length')
        th, tb = self.pack_tail(d['count'], tail)
        # (head[11]: the owner's block stamp, epoch seconds -- the root's result dict must say when the samples arrived, not when
        # the finished block was delivered 2 G - 1 blocks later)
        head = np.array([d['count'], d['doppler'], d['doppler_std'], d['SNR'], d['spSymEst'], len(bits), d['time_ms'] * 1e-3,
                         th[1], th[2], th[3], th[4], float(d['timestamp'])], dtype=np.float64)
        return head, np.concatenate((bits, trust, tb))

    @staticmethod
    def unpack(head, body):
        S = int(head[5])
        _, tail = BlockShard.unpack_tail(np.array([head[0], head[7], head[8], head[9], head[10]]), body[2 * S:])
        return {'count': int(head[0]), 'doppler': float(head[1]), 'doppler_std': float(head[2]), 'SNR': float(head[3]),
                'spSym': float(head[4]), 'spent': float(head[6]), 'timestamp': float(head[11]), 'bits': body[:S],
                'trust': body[S:2 * S], 'tail': tail}

    # -- transport -------------------------------------------------------------------------------------------------------
    TAG_TAIL, TAG_RESULT = 1, 2      # two message kinds may travel between one pair of ranks: matched by tag, not by order

    def _isend(self, head, body, dst, tag):
        torch = self.torch
        th, tb = torch.from_numpy(head), torch.from_numpy(np.ascontiguousarray(body))
        works = [self.dist.isend(th, self.peers[dst], group=self.group, tag=tag)]
        if len(body):
            works.append(self.dist.isend(tb, self.peers[dst], group=self.group, tag=tag))
        self._inflight.append((works, th, tb, dst))
        while len(self._inflight) > 4:
            self._wait_oldest()

    def _wait_oldest(self):
        works, th, _, dst = self._inflight.pop(0)
        self.phase = f'waiting for a send of block {int(th[0])} to rank {dst} to complete'
        for w in works:
            w.wait()
        self.phase = 'between blocks'

    def flush(self):
        while self._inflight:
            self._wait_oldest()

    def _recv(self, src, header_len, body_len, what, tag):
        torch = self.torch
        th = torch.empty(header_len, dtype=torch.float64)
        self.phase = f'recv of {what} from rank {src} (process {self.peers[src]})'
        self.dist.recv(th, self.peers[src], group=self.group, tag=tag)
        head = th.numpy()
        nbytes = body_len(head)
        tb = torch.empty(nbytes, dtype=torch.uint8)
        if nbytes:
            self.dist.recv(tb, self.peers[src], group=self.group, tag=tag)
        self.phase = 'between blocks'
        return head, tb.numpy()

    def send_tail(self, index, tail):
        head, body = self.pack_tail(index, tail)
        self._isend(head, body, self.owner(index + 1), self.TAG_TAIL)

    def recv_tail(self, index):
        head, body = self._recv(self.owner(index), self.TAIL_HEADER, lambda h: int(h[1]) + int(h[2]), f'the tail of block {index}', self.TAG_TAIL)
        got, tail = self.unpack_tail(head, body)
        if got != index:
            raise RuntimeError(f'tail of block {got} arrived where block {index} was expected')
        return tail

    def send_result(self, d, tail):
        head, body = self.pack(d, tail)
        self._isend(head, body, self.root, self.TAG_RESULT)

    def recv_result(self, index):
        head, body = self._recv(self.owner(index), self.HEADER, lambda h: 2 * int(h[5]) + int(h[7]) + int(h[8]),
                                f'the finished block {index}', self.TAG_RESULT)
        if head[0] < 0:                    # (head[0] is the owner's running block counter) -1: the owner failed and says so
            raise RuntimeError(f'rank {self.owner(index)} reported a failure where its finished block {index} was expected')
        return self.unpack(head, body)

    def _poison(self, next_index):
        """This rank is leaving ``run`` with an exception: tell the peers that would otherwise sit in a receive for ever -- the
        owner of the next block (it waits for this rank's tail) and the root (it waits for this rank's finished blocks).  Empty
        messages with block index -1; the receivers raise.  Best effort: nothing here may raise or wait."""
        try:
            torch = self.torch
            if self.world > 1:
                nxt = self.owner(next_index)
                if nxt != self.rank:
                    th = torch.from_numpy(np.array([-1.0, 0, 0, 0, 0]))
                    self._poisoned = [(self.dist.isend(th, self.peers[nxt], group=self.group, tag=self.TAG_TAIL), th)]
                if self.rank != self.root:
                    hh = torch.from_numpy(np.concatenate(([-1.0], np.zeros(self.HEADER - 1))))
                    self._poisoned = getattr(self, '_poisoned', []) + [(self.dist.isend(hh, self.peers[self.root], group=self.group,
                                                                                     tag=self.TAG_RESULT), hh)]
        except Exception:       # noqa: BLE001
            pass

    # -- the loop ------------------------------------------------------------------------------------------------------
    def run(self, runner, sample_source, sink=None, decoder=None, feed=None, skip=None, watchdog=None, feed_begin=None):
        """Drive ``runner`` (a DemodulatorRunner on this rank's device) over the stream of new-sample slices.  Returns
        (results, packets) on the root -- the same as ``DemodulatorRunner.run`` of one process on the whole stream -- and
        ([], []) elsewhere.  ``feed(item) -> part`` / ``skip(item)`` replace ``runner.feed_device`` / ``runner.skip_block``
        for sources that do not yield host sample slices (bench.py: blocks already resident in device memory);
        ``feed_begin(item)`` is the enqueue-only form of ``feed`` (collected with ``runner.feed_device_end``).
        ``watchdog``: a StepWatchdog; it gets one beat per block of the stream.

        An owner enqueues its block on its device (``feed_device_begin``) and collects it when its next block comes up, so
        its host stage runs while its device works; the tail of a block is forwarded the moment its device results are in
        and the next block is known to exist.  The root decodes 2 G - 1 blocks behind the stream (G - 1 with a synchronous
        ``feed``): by then the owner of the block it wants has long sent it."""
        results, packets = [], []
        demod = runner.demod
        G = self.world
        use_async = feed_begin is not None or (feed is None and hasattr(runner, 'feed_device_begin'))
        feed_begin = feed_begin or (lambda c: runner.feed_device_begin(np.asarray(c, dtype=np.complex64)))
        feed = feed or (lambda c: runner.feed_device(np.asarray(c, dtype=np.complex64)))
        skip = skip or runner.skip_block
        is_root = self.rank == self.root
        lag = (2 * G - 1) if use_async else (G - 1)
        state = {'flying': None,         # own block enqueued on the device: its index
                 'tail': None,           # (index, tail) of the own block whose tail waits for block index + 1 to show up
                 'local_tail': None,     # (index, tail) of the last own block: the next block's predecessor when G == 1
                 'seen': -1}             # index of the last block of the stream seen so far
        backlog = []                     # root: blocks to decode, oldest first: (index, finished dict or None = to be received)
        first = [None]

        # where the host time of this run went: the root's work per block (result dict + decoder) apart from its waiting for
        # the block to arrive; an owner's work per own block (tail exchange, bit lookup, alignment, hand-over)
        self.stats = {'root_s': 0.0, 'root_wait_s': 0.0, 'root_blocks': 0, 'host_s': 0.0, 'own_blocks': 0}

        split = decoder is not None and hasattr(decoder, 'findFrames_begin')
        if split and hasattr(decoder, 'prepare'):
            decoder.prepare()
        searching = []                   # the result dict whose bits the decoder is searching right now

        def emit(d):
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
                emit(d)

        def deliver(entry):
            t_in = time.perf_counter()
            i, fin = entry
            if fin is None:
                r = self.recv_result(i)
                self.stats['root_wait_s'] += time.perf_counter() - t_in
                t_in = time.perf_counter()
                d = runner.compose_result(r['count'], r['timestamp'], r['doppler'], r['doppler_std'], r['SNR'], r['bits'], r['trust'],
                                          r['spSym'], r['spent'])
                tail = r['tail']
            else:
                d, tail = fin
            demod.poswinP, demod.posSymEnd = tail['post'], tail['end']          # the root carries the stream's alignment state
            if split:
                # the decoder's two searches of this block run on the device while the root goes on; their hits are collected,
                # and the packet state machine run, right before the next block's bits go in (as run_stream does)
                finish_search()
                decoder.findFrames_begin(d['data'], 0)
                searching.append(d)
            else:
                if decoder is not None:
                    pk, _, nsync = decoder.findFrames(d['data'], 0)
                    d['numSyncSig'] = nsync
                    packets.extend(pk)
                emit(d)
            self.stats['root_s'] += time.perf_counter() - t_in
            self.stats['root_blocks'] += 1

        def finish_own(i, part):
            """Device results of own block i are in: tail out, predecessor's tail in, host stage, hand over."""
            t_in = time.perf_counter()
            tail = demod.overlapTail(part['rec'])
            if not tail['exact']:
                # a window shorter than overlapOffset + 2 symbols: the single-process loop logs its failed alignment and carries
                # on (DB:965-967); so does this one -- the forwarded tail is then the un-adjusted one
                log.error('block %d: window of fewer than overlapOffset + 2 symbols: its tail is forwarded without the +-1 '
                          'adjustment the next block\'s alignment may have needed', i)
            if G > 1 and state['seen'] > i:
                self.send_tail(i, tail)
            else:
                state['tail'] = (i, tail)                      # block i + 1 has not shown up yet (or is this rank's own)
            if i == first[0]:
                prev = None                                    # the stream's first block: the runner's own state
            elif G == 1:
                prev = state['local_tail'][1]
            else:
                prev = self.recv_tail(i - 1)
            state['local_tail'] = (i, tail)
            d = runner.feed_host(part, prev_tail=prev)
            if not is_root:
                self.send_result(d, tail)
            self.stats['host_s'] += time.perf_counter() - t_in
            self.stats['own_blocks'] += 1
            return (d, tail) if is_root else None

        def collect_flying():
            if state['flying'] is None:
                return
            i, state['flying'] = state['flying'], None
            fin = finish_own(i, runner.feed_device_end())
            if is_root:
                for k, (j, q) in enumerate(backlog):
                    if j == i:
                        backlog[k] = (j, fin)

        import time
        try:
            for i, chunk in enumerate(sample_source):
                if first[0] is None:
                    first[0] = i
                state['seen'] = i
                own = self.owner(i) == self.rank
                # a tail that waited for this block to show up
                if state['tail'] is not None and state['tail'][0] == i - 1 and G > 1:
                    self.send_tail(*state['tail'])
                    state['tail'] = None
                if own:
                    collect_flying()                               # this rank's previous block: G blocks back
                    if use_async:
                        feed_begin(chunk)
                        state['flying'] = i
                        if is_root:
                            backlog.append((i, 'flying'))
                    else:
                        fin = finish_own(i, feed(chunk))
                        if is_root:
                            backlog.append((i, fin))
                else:
                    skip(chunk)
                    if is_root:
                        backlog.append((i, None))
                if is_root:
                    while len(backlog) > lag and backlog[0][1] != 'flying':
                        deliver(backlog.pop(0))
                if watchdog is not None:
                    watchdog.beat(i)
            collect_flying()
            state['tail'] = None                                   # the stream's last block has no successor in this call ...
            # ... but the NEXT call's first block (index 0 again: rank 0's) continues the stream: rank 0 must end up with the last
            # block's tail as its alignment state.  The root gets it with the finished block; any other rank 0 gets it from the owner.
            last = state['seen']
            if G > 1 and last >= 0 and self.root != 0 and self.owner(last) != 0:
                if self.owner(last) == self.rank:
                    head, body = self.pack_tail(last, state['local_tail'][1])
                    self._isend(head, body, 0, self.TAG_TAIL)
                elif self.rank == 0:
                    head, body = self._recv(self.owner(last), self.TAIL_HEADER, lambda h: int(h[1]) + int(h[2]),
                                            f'the tail of the last block {last}', self.TAG_TAIL)
                    tail = self.unpack_tail(head, body)[1]
                    demod.poswinP, demod.posSymEnd = tail['post'], tail['end']
            while backlog:
                deliver(backlog.pop(0))
                if watchdog is not None:
                    watchdog.beat()         # (the drain is 2 G - 1 blocks of decoder work on the root)
            if is_root:
                t_in = time.perf_counter()
                finish_search()
                self.stats['root_s'] += time.perf_counter() - t_in
            self.flush()
            return results, packets

        except BaseException:
            # leave nothing behind a failure: no block in flight on this rank's device (the handle must be usable afterwards),
            # no peer waiting for a message that will never come
            if state['flying'] is not None:
                state['flying'] = None
                try:
                    runner.feed_device_end()
                except Exception:       # noqa: BLE001
                    pass
            if searching:
                try:
                    decoder.findFrames_end()
                except Exception:       # noqa: BLE001
                    pass
            self._poison(state['seen'] + 1)
            raise
        finally:
            if watchdog is not None:
                watchdog.suspend()      # the caller owns stop(); an idle or busy caller is not "no progress"


# ---- Doppler bins x time chunks ----------------------------------------------------------------------------------------
class GridShard:
    """The two axes together ("two independent axes"): the job's G processes form T = G / B groups of B ranks.
    Inside a group the Doppler bins are sharded (DopplerShard on the group's communicator: block broadcast from the group's
    first rank, one collective on the scores, pick and demodulation on every rank); the groups take the blocks round-robin
    (BlockShard between the groups' first ranks: one point-to-point message per block back to process 0, which runs the
    sequential host stages and the decoder in block order).  Process rank = group * B + bin rank.

        grid = GridShard(bin_ranks=4)                        # 8 processes: 2 groups x 4 bin slices
        runner = DemodulatorRunner(conf, protocol, radio, shard=grid.doppler)
        results, packets = grid.run(runner, chunks, decoder=Decoder(conf, protocol))

    ``B = 1`` is BlockShard, ``B = G`` is DopplerShard with the streaming loop around it.  ``hand_back_backend``: backend
    of the communicator that moves the hand-back (host arrays): gloo."""

    def __init__(self, bin_ranks, device=None, comm=None, hand_back_backend='gloo', concurrent_broadcast=True):
        if comm is None:
            import torch.distributed as comm
        world, rank = comm.get_world_size(), comm.get_rank()
        B = int(bin_ranks)
        if B < 1 or world % B:
            raise ValueError(f'{world} processes do not divide into groups of {bin_ranks}')
        self.B, self.T = B, world // B
        self.g, self.b = divmod(rank, B)
        # every process creates every communicator, in the same order (torch.distributed's rule for new_group)
        bin_groups = [comm.new_group(ranks=list(range(g * B, (g + 1) * B))) for g in range(self.T)]
        bcast_groups = ([comm.new_group(ranks=list(range(g * B, (g + 1) * B))) for g in range(self.T)]
                        if B > 1 and concurrent_broadcast else bin_groups)
        roots = [g * B for g in range(self.T)]
        root_group = comm.new_group(ranks=roots, backend=hand_back_backend)
        self.doppler = DopplerShard(rank=self.b, world=B, group=bin_groups[self.g], device=device, comm=comm, src=self.g * B,
                                    bcast_group=bcast_groups[self.g], concurrent_broadcast=concurrent_broadcast)
        self.blocks = BlockShard(rank=self.g, world=self.T, group=root_group, comm=comm, ranks=roots) if self.b == 0 else None

    def owner_group(self, block_index):
        return int(block_index) % self.T

    def describe(self):
        return self.doppler.describe() + ('' if self.blocks is None else '; ' + self.blocks.describe())

    def run(self, runner, sample_source, sink=None, decoder=None, watchdog=None):
        """``runner``: a DemodulatorRunner built with ``shard=self.doppler``.  Every process iterates the same stream of
        new-sample slices (only the first rank of a group reads the samples of the blocks its group owns; the others take
        them from its broadcast).  Returns (results, packets) on process 0, ([], []) elsewhere."""
        if self.blocks is not None:
            return self.blocks.run(runner, sample_source, sink=sink, decoder=decoder, watchdog=watchdog)
        try:
            for i, chunk in enumerate(sample_source):
                if self.owner_group(i) == self.g:
                    runner.feed_device(np.asarray(chunk, dtype=np.complex64))
                else:
                    runner.skip_block(chunk)
                if watchdog is not None:
                    watchdog.beat(i)
        finally:
            if watchdog is not None:
                watchdog.suspend()          # the caller owns stop()
        return [], []
