"""Rank of the Doppler-bins x time-chunks test (launched through torch.distributed.run): groups of ``bin_ranks``
processes shard the Doppler bins of a block, the groups take the blocks round-robin, process 0 runs the sequential host
stages and the decoder.  The result must equal one process with the whole bin table on the whole stream.
argv: backend bin_ranks [log2N [bins]].  With gloo the ranks share the device (rehearsal on a 1-GPU box).  Prints one JSON
line per rank."""
import json
import datetime
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
from pycusdr_amd.dist import GridShard, StepWatchdog         # noqa: E402
from pycusdr_amd.protocol import loadProtocol  # noqa: E402

rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
backend = sys.argv[1] if len(sys.argv) > 1 else 'gloo'
bin_ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 15
D = int(sys.argv[4]) if len(sys.argv) > 4 else 64
if backend == 'gloo':
    local = local % torch.cuda.device_count()
torch.cuda.set_device(local)
dist.init_process_group(backend, device_id=torch.device('cuda', local) if backend == 'nccl' else None,
                        timeout=datetime.timedelta(seconds=90))     # peers that never arrive cost 90 s, not ten minutes
N, ov = 1 << bs, 1 << 10
conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=D, device=local)
p = loadProtocol('bench_GMSK')(conf=conf)
pkt = sg.get_padded_packet('GMSK', 16, 153600)[0]
sig = sg.awgn(np.concatenate((pkt, np.zeros(2 * N))), 12.0, rng=np.random.RandomState(7)).astype(np.complex64)
step = N - ov
nblocks = (len(sig) - ov) // step
chunks = [sig[ov + i * step: ov + (i + 1) * step] for i in range(nblocks)]
grid = GridShard(bin_ranks, device=torch.device('cuda', local))
run = DemodulatorRunner(conf, p, 'UHF-H', shard=grid.doppler)
run.raw[:ov] = sig[:ov]
dog = StepWatchdog(120.0, rank=rank, describe=grid.describe)      # a collective nobody joins: diagnosis + exit 3
res, packets = grid.run(run, chunks, decoder=Decoder(conf, p), watchdog=dog)
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
ok = ok and run.demod.bank.D == len(range(*grid.doppler.bin_range(D)))
print(json.dumps({'rank': rank, 'ok': bool(ok), 'failed': ','.join(why), 'blocks': nblocks, 'group': grid.g, 'bin_rank': grid.b}), flush=True)
run.close()
dist.barrier()
dist.destroy_process_group()
