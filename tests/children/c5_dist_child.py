"""Rank of the C5 multi-GPU rehearsal (BASELINE config 5: two concurrent demodulator instances -- CC11xx FSK-2 and the custom
BPSK filter set -- each with its bins sharded over all ranks): every process holds BOTH instances, each on a communicator of
its own (the two instances time-share the GPUs), blocks of the two streams alternate.  Each instance's sharded pick and table
must equal its unsharded handle's bit for bit, and every rank must demodulate the same bits.  argv: backend log2N bins.
Prints one JSON line per rank."""
import json
import datetime
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch                                   # noqa: E402
import torch.distributed as dist               # noqa: E402
from c5_common import c5_instance              # noqa: E402
from pycusdr_amd.demodulator import UHF        # noqa: E402
from pycusdr_amd.dist import DopplerShard, StepWatchdog      # noqa: E402

rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
backend = sys.argv[1] if len(sys.argv) > 1 else 'gloo'
log2N = int(sys.argv[2]) if len(sys.argv) > 2 else 17
D = int(sys.argv[3]) if len(sys.argv) > 3 else 64
if backend == 'gloo':
    local = local % torch.cuda.device_count()
torch.cuda.set_device(local)
dist.init_process_group(backend, device_id=torch.device('cuda', local) if backend == 'nccl' else None,
                        timeout=datetime.timedelta(seconds=90))     # peers that never arrive cost 90 s, not ten minutes
dev = torch.device('cuda', local)
ok, failed = True, []
inst, sharded, plain = {}, {}, {}
for name in ('CC11xx', 'bench_BPSK'):
    inst[name] = c5_instance(name, log2N, D)
    conf = inst[name]['conf']
    conf['GPU']['UHF']['CUDA']['device'] = local
    group = dist.new_group(backend=backend)                    # one communicator per instance
    sharded[name] = UHF.Demodulator(conf, inst[name]['proto'], 'UHF-H', shard=DopplerShard(group=group, device=dev))
    if rank == 0:
        plain[name] = UHF.Demodulator(conf, inst[name]['proto'], 'UHF-H')
dog = StepWatchdog(180.0, rank=rank, describe=lambda: ' | '.join(f'{n}: {d.shard.describe()}' for n, d in sharded.items()))
for rep in range(2):
    for name in ('CC11xx', 'bench_BPSK'):                      # the two streams alternate on the same devices
        dog.beat()
        x = inst[name]['x']
        res = sharded[name].uploadAndFindCarrier(x if rank == 0 else None)
        full = sharded[name].shard.full_scores()
        out = sharded[name].demodulate()
        if rank == 0:
            ref = plain[name].uploadAndFindCarrier(x.copy())
            pr = plain[name].demodulate()
            checks = dict(estimate=bool(np.array_equal([res[0], res[1], res[3]], [ref[0], ref[1], ref[3]], equal_nan=True)),
                          table=bool(np.array_equal(full, plain[name].bank.get_scores())),
                          symbols=all(np.array_equal(u, v) for u, v in zip(out[:3], pr[:3])) and out[3] == pr[3],
                          carrier=abs(int(plain[name].dopplerIdxlast) - inst[name]['expect']) <= 2 * np.median(np.abs(np.diff(inst[name]['shifts'].astype(np.int64)))))
            failed += [f'{name}/{rep}: {k}' for k, v in checks.items() if not v]
            ok &= all(checks.values())
        h = float(zlib.crc32(out[0].tobytes() + out[1].tobytes() + out[2].tobytes()))
        t = torch.tensor([float(res[0]), float(res[1]), h, float(out[3])], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        g = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(g, t)
        ok &= all(bool(torch.equal(g[0], q)) for q in g)
dog.stop()
print(json.dumps({'rank': rank, 'ok': bool(ok), 'failed': '; '.join(failed)}), flush=True)
for d in list(sharded.values()) + list(plain.values()):
    d.close()
dist.barrier()
dist.destroy_process_group()
