"""The sharded (RCCL) search path on one GPU: a 1-rank nccl process group exercises DopplerShard
(side stream, export, all-reduce, pick on the reduced matrix) and must equal the unsharded result."""
import os
import socket

import numpy as np
import pytest

from pycusdr_amd import config as cfg, signals as sg
from pycusdr_amd.protocol import loadProtocol

pytestmark = pytest.mark.gpu


def test_sharded_search_equals_local_search_one_rank():
    import torch
    import torch.distributed as dist
    from pycusdr_amd.demodulator import UHF
    from pycusdr_amd.dist import DopplerShard
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        bs = 14
        N = 1 << bs
        conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=24)
        p = loadProtocol('bench_GMSK')(conf=conf)
        plain = UHF.Demodulator(conf, p, 'UHF-H')
        sharded = UHF.Demodulator(conf, p, 'UHF-H', shard=DopplerShard())
        x = sg.get_padded_packet('GMSK')[0][30000:30000 + N].astype(np.complex64)
        for _ in range(3):
            a = plain.uploadAndFindCarrier(x)
            b = sharded.uploadAndFindCarrier(x)
            assert a[0] == b[0] and a[1] == b[1] and a[3] == b[3]
            assert int(plain.dopplerIdxlast) == int(sharded.dopplerIdxlast) == N // 4
            full = sharded.shard.full_scores()
            assert np.array_equal(full, plain.bank.get_scores())
            ra, rb = plain.demodulate(), sharded.demodulate()
            assert all(np.array_equal(u, v) for u, v in zip(ra[:3], rb[:3])) and ra[3] == rb[3]
        plain.close()
        sharded.close()
    finally:
        dist.destroy_process_group()
