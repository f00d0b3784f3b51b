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
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist
    from oracle import mfbank_oracle as orc
    from pycusdr_amd.dist import allreduce_scores_host, bin_slice as bs
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rs = np.random.RandomState(5)              # same block and filters on every rank
    N, M, D = 1 << 12, 4, 10
    x = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
    masks = (rs.standard_normal((M, N)) + 1j * rs.standard_normal((M, N))).astype(np.complex64)
    shifts = rs.randint(0, N, D)
    X = orc.forward_fft(x)
    lo, hi = bs(D, rank, world)
    local = orc.doppler_scores(X, masks, shifts[lo:hi], True).astype(np.float32)
    full = allreduce_scores_host(local, lo, D)
    idx, metric = orc.find_doppler_est(full, D, 0, True)
    single = orc.doppler_scores(X, masks, shifts, True).astype(np.float32)
    sidx, smetric = orc.find_doppler_est(single, D, 0, True)
    ok = bool(np.array_equal(full, single) and idx == sidx and metric == smetric)   # adding exact zeros
    q.put((rank, ok, float(idx)))
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
