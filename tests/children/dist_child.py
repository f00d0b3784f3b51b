"""Rank of the multi-GPU parity test (launched through torch.distributed.run, one rank per GPU): the
sharded Doppler search + RCCL exchange must pick, bit for bit, what one unsharded handle picks on the full
bin table.  Prints one JSON line per rank."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch                                   # noqa: E402
import torch.distributed as dist               # noqa: E402
from pycusdr_amd import config as cfg, signals as sg   # noqa: E402
from pycusdr_amd.demodulator import UHF        # noqa: E402
from pycusdr_amd.dist import DopplerShard      # noqa: E402
from pycusdr_amd.protocol import loadProtocol  # noqa: E402

rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
backend = sys.argv[1] if len(sys.argv) > 1 else 'nccl'
if backend == 'gloo':                      # rehearsal on a 1-GPU box: the ranks share the device
    local = local % torch.cuda.device_count()
torch.cuda.set_device(local)
dist.init_process_group(backend, device_id=torch.device('cuda', local) if backend == 'nccl' else None)
bs, D = 16, 64 * world
N = 1 << bs
conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=D, device=local)
p = loadProtocol('bench_GMSK')(conf=conf)
sharded = UHF.Demodulator(conf, p, 'UHF-H', shard=DopplerShard(device=torch.device('cuda', local)))
plain = UHF.Demodulator(conf, p, 'UHF-H') if rank == 0 else None
sig = sg.s1_stream(3, N, 1 << 10, 'GMSK', snr_db=12.0, seed=5)
ok = True
for b in range(3):
    x = sig[b * (N - 1024): b * (N - 1024) + N]
    res = sharded.uploadAndFindCarrier(x if rank == 0 else None)     # only rank 0 owns the stream
    full = sharded.shard.full_scores()
    out = sharded.demodulate()
    owner = sharded.shard.owner(sharded._pick_bin)
    if rank == 0:
        ref = plain.uploadAndFindCarrier(x)
        ok &= res[0] == ref[0] and res[1] == ref[1] and res[3] == ref[3]
        ok &= bool(np.array_equal(full, plain.bank.get_scores()))
        ok &= int(sharded.dopplerIdxlast) == int(plain.dopplerIdxlast) == N // 4
        if owner == 0:
            pr = plain.demodulate()
            ok &= all(np.array_equal(u, v) for u, v in zip(out[:3], pr[:3]))
    ok &= (len(out[0]) > 0) == (owner == rank)        # the demodulation stage runs on the owner only
    t = torch.tensor([float(res[0]), float(res[1])], device='cuda' if backend == 'nccl' else 'cpu')
    g = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(g, t)
    ok &= all(bool(torch.equal(g[0], q)) for q in g)  # every rank picked the same
print(json.dumps({'rank': rank, 'ok': bool(ok)}), flush=True)
sharded.close()
if plain is not None:
    plain.close()
dist.barrier()
dist.destroy_process_group()
