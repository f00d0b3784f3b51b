"""One rank of the watchdog test (CPU, gloo, oracle bank): a stream of blocks through DopplerShard with a StepWatchdog
beside it.  argv: rank world port withheld_rank stall_at_step single_comm.  The withheld rank stops joining the collectives at
``stall_at_step`` (it sleeps); every other rank must notice within its watchdog's timeout, say which collective it is
stuck in, and leave with status 3."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch                                   # noqa: E402
import torch.distributed as dist               # noqa: E402
from oracle_bank import OracleBank             # noqa: E402
from pycusdr_amd.dist import DopplerShard, StepWatchdog      # noqa: E402

rank, world, port, withheld, stall_at, single = (int(v) for v in sys.argv[1:7])
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
dist.init_process_group('gloo', rank=rank, world_size=world)
rs = np.random.RandomState(5)
log2N, M, D = 10, 2, 8
N = 1 << log2N
masks = (rs.standard_normal((M, N)) + 1j * rs.standard_normal((M, N))).astype(np.complex64)
shifts = rs.randint(0, N, D)
shard = DopplerShard(device=torch.device('cpu'), concurrent_broadcast=not single)
lo, hi = shard.bin_range(D)
bank = OracleBank(log2N, hi - lo, M, sum_all_masks=True)
bank.set_filters(masks)
bank.set_shifts(shifts[lo:hi])
shard.attach(bank, D, M, sum_all=True)
dog = StepWatchdog(2.0, rank=rank, describe=shard.describe, first_grace_s=20.0)
x = (rs.standard_normal(N) + 1j * rs.standard_normal(N)).astype(np.complex64)
for i in range(6):
    if rank == withheld and i == stall_at:
        dog.stop()
        time.sleep(30)              # never joins step `stall_at`; the test kills this process
        sys.exit(0)
    block = torch.from_numpy(x.view(np.float32).copy()) if rank == 0 else None
    shard.step(bank, lo, block)
    dog.beat(i)
dog.stop()
print(f'rank {rank}: six steps done', flush=True)
dist.destroy_process_group()
