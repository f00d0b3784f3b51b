"""Multi-GPU Doppler-bin sharding: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" on CPU for tests).

The reference has no distributed runtime.  The hot path shards by Doppler bin:
rank r of G owns the contiguous slice [r*D/G, (r+1)*D/G) of the bin table and holds a full replica of
the filter bank.  Per block:
  1. rank 0 owns the IQ stream; its N-sample block is broadcast to every rank (8 MiB over xGMI, on the
     shard's side stream) -- ranks need no feeder process of their own;
  2. every rank searches its bins (libmfbank, same stream);
  3. ONE all-reduce (sum) of the per-bin scores in which every rank has zeros outside its slice --
     adding exact zeros, so the result is bit-identical on all ranks.  With SUM_ALL_MASKS (every shipped
     protocol) only column 0 of doppSum is populated (reference cuda_kernels.cu:453-464), so only that
     column travels: D floats instead of D*M;
  4. the Doppler pick on the full table on every rank (identical everywhere);
  5. the demodulation stage (1/D of the work) runs on the rank that owns the picked bin only.
The exchange is latency-bound (<= 8 KiB), not link-bandwidth-bound.
"""
import contextlib

import numpy as np


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


class DopplerShard:
    """Glue between a bank (MFBank, or any object with its device-pointer methods) holding this rank's
    bins and the process group.  Works on CUDA/HIP tensors over RCCL and on CPU tensors over gloo."""

    def __init__(self, rank=None, world=None, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
        self.device = torch.device(device)
        self.on_gpu = self.device.type == 'cuda'
        self.scores = None
        self.block = None
        # a dedicated side stream: the bank's kernels, the buffer clear and the collectives are all
        # ordered on it (an RCCL op is synchronised against the *current* torch stream)
        self.stream = torch.cuda.Stream(self.device) if self.on_gpu else None

    def _on_stream(self):
        return self.torch.cuda.stream(self.stream) if self.on_gpu else contextlib.nullcontext()

    def bin_range(self, num_bins):
        return bin_slice(num_bins, self.rank, self.world)

    def owner(self, bin_index):
        return bin_owner(self.D, self.world, int(bin_index))

    def attach(self, bank, num_bins_total, M, sum_all=False):
        """Allocate the exchange buffers and run the bank on the shard's stream so that the collectives
        are ordered after the search without host synchronisation."""
        torch = self.torch
        self.D, self.M, self.sum_all = int(num_bins_total), int(M), bool(sum_all)
        shape = (self.D,) if self.sum_all else (self.D, self.M)
        self.scores = torch.zeros(shape, dtype=torch.float32, device=self.device)
        self.block = torch.empty(2 * bank.N, dtype=torch.float32, device=self.device)   # complex64 block, interleaved
        if self.on_gpu:
            torch.cuda.synchronize(self.device)
            bank.set_stream(self.stream.cuda_stream)

    def full_scores(self):
        """The reduced score table as float32 [D, M] (column 0 only is populated under SUM_ALL_MASKS)."""
        s = self.scores.detach().cpu().numpy()
        if not self.sum_all:
            return s
        out = np.zeros((self.D, self.M), dtype=np.float32)
        out[:, 0] = s
        return out

    def broadcast_block(self, bank, block=None):
        """Rank 0 passes the block (a float32 view of N complex64 samples, on the shard's device); every rank
        receives it and runs the forward FFT on it."""
        with self._on_stream():
            if self.rank == 0:
                if block is None:
                    raise ValueError('rank 0 must supply the block')
                buf = block if block.is_contiguous() else block.contiguous()
            else:
                buf = self.block
            if self.world > 1:
                self.dist.broadcast(buf, src=0, group=self.group)
            bank.upload_device(buf.data_ptr())
            self._live = buf      # keep the source alive until the next block

    def search_and_pick(self, bank, row_offset):
        with self._on_stream():
            self.scores.zero_()
            bank.search_async()
            if self.sum_all:
                bank.export_column_async(self.scores.data_ptr(), row_offset)
            else:
                bank.export_scores_async(self.scores.data_ptr(), row_offset)
            self.dist.all_reduce(self.scores, op=self.dist.ReduceOp.SUM, group=self.group)
            if self.sum_all:
                return bank.pick_column(self.scores.data_ptr(), num=self.D, offset=0)
            return bank.pick(self.scores.data_ptr(), num=self.D, offset=0)

    def step(self, bank, row_offset, block=None):
        """One block of the sharded hot path: broadcast, search, exchange, pick."""
        self.broadcast_block(bank, block)
        return self.search_and_pick(bank, row_offset)


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
