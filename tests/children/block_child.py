"""Rank of the time-chunk sharding test (launched through torch.distributed.run, ranks may share the device): rank r
runs the device stages of blocks r, r + G, ... on its own library handle; rank 0 runs the sequential host stages and the
decoder in block order.  The result must equal one process running the whole stream.  Prints one JSON line per rank."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch                                   # noqa: E402
import torch.distributed as dist               # noqa: E402
from pycusdr_amd import config as cfg, signals as sg   # noqa: E402
from pycusdr_amd.decoder import Decoder        # noqa: E402
from pycusdr_amd.demodulator_process import DemodulatorRunner   # noqa: E402
from pycusdr_amd.dist import BlockShard, StepWatchdog        # noqa: E402
from pycusdr_amd.protocol import loadProtocol  # noqa: E402

rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 15
local = local % torch.cuda.device_count()
torch.cuda.set_device(local)
dist.init_process_group('gloo')
N, ov = 1 << bs, 1 << 10
conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=64, device=local)
p = loadProtocol('bench_GMSK')(conf=conf)
pkt = sg.get_padded_packet('GMSK', 16, 153600)[0]
sig = sg.awgn(np.concatenate((pkt, np.zeros(2 * N))), 12.0, rng=np.random.RandomState(7)).astype(np.complex64)
step = N - ov
nblocks = (len(sig) - ov) // step
chunks = [sig[ov + i * step: ov + (i + 1) * step] for i in range(nblocks)]
run = DemodulatorRunner(conf, p, 'UHF-H')
run.raw[:ov] = sig[:ov]
shard = BlockShard()
dog = StepWatchdog(120.0, rank=rank, describe=shard.describe)       # a hand-back that never arrives: diagnosis + exit 3
res, packets = shard.run(run, chunks, decoder=Decoder(conf, p), watchdog=dog)
dog.stop()
ok, why = True, []
if rank == 0:
    plain = DemodulatorRunner(conf, p, 'UHF-H')
    plain.raw[:ov] = sig[:ov]
    ref, ref_packets = plain.run(chunks, decoder=Decoder(conf, p))
    checks = {'blocks': len(res) == len(ref) == nblocks,
              'estimates': all(np.array_equal([a['doppler'], a['doppler_std'], a['SNR'], a['spSymEst']],
                                              [b['doppler'], b['doppler_std'], b['SNR'], b['spSymEst']], equal_nan=True)
                               for a, b in zip(res, ref)),
              'bits': all(np.array_equal(a['data'], b['data']) and np.array_equal(a['trust'], b['trust']) for a, b in zip(res, ref)),
              'state': bool(np.array_equal(run.demod.poswinP, plain.demod.poswinP) and np.array_equal(run.demod.posSymEnd, plain.demod.posSymEnd)),
              'packets': len(packets) == len(ref_packets) == 1 and bool(np.array_equal(packets[0].bits, ref_packets[0].bits)),
              'no_bit_errors': len(packets) == 1 and packets[0].checkPacketData() == 0}
    why = [k for k, v in checks.items() if not v]
    ok = not why
    plain.close()
else:
    ok = res == [] and packets == [] and run.count == nblocks
print(json.dumps({'rank': rank, 'ok': bool(ok), 'failed': ','.join(why), 'blocks': nblocks}), flush=True)
run.close()
dist.barrier()
dist.destroy_process_group()
