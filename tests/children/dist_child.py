"""Rank of the multi-rank parity tests (launched through torch.distributed.run): the sharded Doppler search + exchange
must pick, bit for bit, what one unsharded handle picks on the full bin table, and every rank must demodulate the same
bits.  argv: backend [log2N [bins_total [noise_bin]]].  Prints one JSON line per rank."""
import json
import datetime
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch                                   # noqa: E402
import torch.distributed as dist               # noqa: E402
from pycusdr_amd import config as cfg, signals as sg   # noqa: E402
from pycusdr_amd.demodulator import UHF        # noqa: E402
from pycusdr_amd.dist import DopplerShard, StepWatchdog      # noqa: E402
from pycusdr_amd.protocol import loadProtocol  # noqa: E402

rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
backend = sys.argv[1] if len(sys.argv) > 1 else 'nccl'
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
D = int(sys.argv[3]) if len(sys.argv) > 3 else 64 * world
noise = int(sys.argv[4]) if len(sys.argv) > 4 else 0
if backend == 'gloo':                      # rehearsal on a 1-GPU box: the ranks share the device
    local = local % torch.cuda.device_count()
torch.cuda.set_device(local)
dist.init_process_group(backend, device_id=torch.device('cuda', local) if backend == 'nccl' else None,
                        timeout=datetime.timedelta(seconds=90))     # peers that never arrive cost 90 s, not ten minutes
N = 1 << bs
conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=D, device=local)
if noise:
    conf['Radios']['Rx']['UHF-H']['noise_measure_offset_Hz'] = 60000      # the reference's noise-reference bin (DB:148-159)
p = loadProtocol('bench_GMSK')(conf=conf)
sharded = UHF.Demodulator(conf, p, 'UHF-H', shard=DopplerShard(device=torch.device('cuda', local)))
plain = UHF.Demodulator(conf, p, 'UHF-H') if rank == 0 else None
nblk = 3
spacing = float(np.median(np.abs(np.diff(sharded.doppCyperSymNorm[noise:].astype(np.int64)))))     # one bin of the search grid
sig = sg.s1_stream(nblk, N, 1 << 10, 'GMSK', snr_db=12.0, seed=5)
ok = True
failed = []
dog = StepWatchdog(120.0, rank=rank, describe=sharded.shard.describe)     # a collective nobody joins: diagnosis + exit 3
for b in range(nblk):
    dog.beat(b)
    x = sig[b * (N - 1024): b * (N - 1024) + N]
    res = sharded.uploadAndFindCarrier(x if rank == 0 else None)     # only rank 0 owns the stream
    full = sharded.shard.full_scores()
    out = sharded.demodulate()
    if rank == 0:
        ref = plain.uploadAndFindCarrier(x)
        pr = plain.demodulate()
        checks = dict(estimate=bool(np.array_equal([res[0], res[1], res[3]], [ref[0], ref[1], ref[3]], equal_nan=True)),
                      table=bool(np.array_equal(full, plain.bank.get_scores())),
                      shift=int(sharded.dopplerIdxlast) == int(plain.dopplerIdxlast) and abs(int(plain.dopplerIdxlast) - N // 4) <= spacing,
                      symbols=all(np.array_equal(u, v) for u, v in zip(out[:3], pr[:3])) and out[3] == pr[3])
        failed += [f'block {b}: {k}' for k, v in checks.items() if not v]
        ok &= all(checks.values())
    ok &= len(out[0]) > N // 16 - 200        # the demodulation stage runs on every rank ...
    h = float(zlib.crc32(out[0].tobytes() + out[1].tobytes() + out[2].tobytes()))
    t = torch.tensor([float(res[0]), float(res[1]), h, float(out[3])], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
    g = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(g, t)
    ok &= all(bool(torch.equal(g[0], q)) for q in g)  # ... and every rank picked and demodulated the same
dog.stop()
print(json.dumps({'rank': rank, 'ok': bool(ok), 'failed': '; '.join(failed), 'even': bool(sharded.shard.even), 'noise_rows': int(sharded.doppIdxArrayOffset)}), flush=True)
sharded.close()
if plain is not None:
    plain.close()
dist.barrier()
dist.destroy_process_group()
