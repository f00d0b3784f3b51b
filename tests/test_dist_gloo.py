"""Multi-rank path on CPU: world_size-2 gloo processes shard the Doppler bins, each computes its
slice of the score matrix (oracle), all-reduce, and every rank must pick exactly what a single
process picks on the full table."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from pycusdr_amd.dist import bin_slice


def test_bin_slice_partitions():
    for D in (1, 7, 256, 2048, 1001):
        for G in (1, 2, 3, 8):
            parts = [bin_slice(D, r, G) for r in range(G)]
            assert parts[0][0] == 0 and parts[-1][1] == D
            assert all(parts[i][1] == parts[i + 1][0] for i in range(G - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    """One rank: DopplerShard itself (CPU tensors over gloo) around the OracleBank stand-in: rank 0 broadcasts the
    block, every rank searches its bin slice, one all-reduce, the pick on every rank."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    import torch
    import torch.distributed as dist
    from oracle import mfbank_oracle as orc
    from oracle_bank import OracleBank
    from pycusdr_amd.dist import DopplerShard, allreduce_scores_host, bin_owner
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rs = np.random.RandomState(5)              # same filters on every rank; the block lives on rank 0 only
    log2N, M, D = 12, 4, 10
    N = 1 << log2N
    x = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
    masks = (rs.standard_normal((M, N)) + 1j * rs.standard_normal((M, N))).astype(np.complex64)
    shifts = rs.randint(0, N, D)
    X = orc.forward_fft(x)
    ok = True
    for sum_all, D in ((True, 10), (False, 10), (True, 7), (False, 7)):    # even slices: all-gather; uneven: all-reduce
        shifts = rs.randint(0, N, D)
        shard = DopplerShard(device=torch.device('cpu'))
        lo, hi = shard.bin_range(D)
        bank = OracleBank(log2N, hi - lo, M, sum_all_masks=sum_all)
        bank.set_filters(masks)
        bank.set_shifts(shifts[lo:hi])
        shard.attach(bank, D, M, sum_all=sum_all)
        block = torch.from_numpy(x.view(np.float32).copy()) if rank == 0 else None
        idx, metric = shard.step(bank, lo, block)
        single = orc.doppler_scores(X, masks, shifts, sum_all).astype(np.float32)
        sidx, smetric = orc.find_doppler_est(single, D, 0, sum_all)
        ok &= bool(np.array_equal(shard.full_scores(), single))       # adding exact zeros
        ok &= bool(idx == sidx and metric == smetric)
        ok &= shard.owner(int(float(idx))) == bin_owner(D, world, int(float(idx)))
        if sum_all:
            ok &= shard.scores.numel() == D                            # only the populated column travels
        # a stream of blocks: the next block is distributed while the current one is searched
        b1 = torch.from_numpy(x.view(np.float32).copy()) if rank == 0 else None
        b2 = torch.from_numpy((x * np.complex64(2)).view(np.float32).copy()) if rank == 0 else None
        i1, _ = shard.step(bank, lo, b1, next_block=b2, prefetch_next=True)
        ok &= bool(np.array_equal(shard.full_scores(), single))
        i2, _ = shard.step(bank, lo, None, next_block=b1, prefetch_next=True)      # block 2 is already there
        ok &= bool(np.array_equal(shard.full_scores(), single * np.float32(4))) and i1 == sidx and i2 == sidx
        i3, _ = shard.step(bank, lo, None)                                         # and block 1 again
        ok &= bool(np.array_equal(shard.full_scores(), single)) and i3 == sidx
    # the documented numpy statement of the same exchange
    local = orc.doppler_scores(X, masks, shifts[lo:hi], True).astype(np.float32)
    ok &= bool(np.array_equal(allreduce_scores_host(local, lo, D), orc.doppler_scores(X, masks, shifts, True).astype(np.float32)))
    q.put((rank, bool(ok), float(idx)))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sharded_scores_allreduce_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == res[1][2]
