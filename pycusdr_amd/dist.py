"""Multi-GPU Doppler-bin sharding: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" on CPU for tests).

The reference has no distributed runtime.  The hot path shards by Doppler bin:
rank r of G owns the contiguous slice [r*D/G, (r+1)*D/G) of the bin table, holds a full replica of
the filter bank and sees the same IQ block.  The only exchange per block is one all-reduce (sum)
of the float32 score matrix [D, M] in which every rank has zeros outside its slice -- adding exact
zeros, so the result is bit-identical on all ranks -- followed by the Doppler pick on the full
matrix on every rank.  It is latency-bound (<= 64 KiB), not link-bandwidth-bound.
"""
import numpy as np


def bin_slice(num_bins, rank, world):
    """Contiguous, near-equal partition of ``num_bins`` over ``world`` ranks."""
    base, rem = divmod(num_bins, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class DopplerShard:
    """Glue between an MFBank holding this rank's bins and the process group."""

    def __init__(self, rank=None, world=None, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.device = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.scores = None
        # a dedicated side stream: the bank's kernels, the buffer clear and the collective are all
        # ordered on it (the RCCL op is synchronised against the *current* torch stream)
        self.stream = torch.cuda.Stream(self.device)

    def bin_range(self, num_bins):
        return bin_slice(num_bins, self.rank, self.world)

    def attach(self, bank, num_bins_total, M):
        """Allocate the [D_total, M] score buffer and run the bank on torch's current stream so the
        collective is ordered after the search without host synchronisation."""
        torch = self.torch
        self.D, self.M = int(num_bins_total), int(M)
        self.scores = torch.zeros((self.D, self.M), dtype=torch.float32, device=self.device)
        torch.cuda.synchronize(self.device)
        bank.set_stream(self.stream.cuda_stream)

    def search_and_pick(self, bank, row_offset):
        with self.torch.cuda.stream(self.stream):
            self.scores.zero_()
            bank.search_async()
            bank.export_scores_async(self.scores.data_ptr(), row_offset)
            self.dist.all_reduce(self.scores, op=self.dist.ReduceOp.SUM, group=self.group)
            return bank.pick(self.scores.data_ptr(), num=self.D, offset=0)


def allreduce_scores_host(local_scores, row_offset, num_bins_total, group=None):
    """CPU/gloo statement of the same exchange on numpy arrays (used by the world_size-2 tests and as
    documentation of the collective): returns the full [D_total, M] matrix on every rank."""
    import torch
    import torch.distributed as dist
    local_scores = np.asarray(local_scores, dtype=np.float32)
    full = torch.zeros((num_bins_total, local_scores.shape[1]), dtype=torch.float32)
    full[row_offset:row_offset + local_scores.shape[0]] = torch.from_numpy(local_scores)
    dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
    return full.numpy()
