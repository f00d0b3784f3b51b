"""Where the host time of a C2 block goes (one process, blocks resident in HBM): device call, tail + bit lookup + alignment
(the owner's host stage under time-chunk sharding), result dict, decoder (the root's share).  usage: python tools/host_shares.py [blocks]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
from pycusdr_amd.hostcpu import quiet_blas  # noqa: E402
quiet_blas()
import torch  # noqa: E402
from pycusdr_amd import config as cfg, signals as sg  # noqa: E402
from pycusdr_amd.decoder import Decoder  # noqa: E402
from pycusdr_amd.demodulator_process import DemodulatorRunner  # noqa: E402
from pycusdr_amd.protocol import loadProtocol  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 60
log2N, ov = 20, 1 << 10
N = 1 << log2N
conf = cfg.bench_config('bench_GMSK', blockSize=log2N, doppCarrierSteps=256)
proto = loadProtocol('bench_GMSK')(conf=conf)
run = DemodulatorRunner(conf, proto, 'UHF-H')
dec = Decoder(conf, proto)
dec.prepare()
stream = sg.s1_stream(16, N, ov, 'GMSK', 16, 153600, snr_db=10.0, seed=1)
host = np.stack([stream[b * (N - ov): b * (N - ov) + N] for b in range(16)])
blocks = torch.from_numpy(host.view(np.float32).reshape(16, 2 * N)).cuda()
esz = blocks.element_size() * 2 * N
torch.cuda.synchronize()
t = {'device': 0.0, 'tail': 0.0, 'host_stage': 0.0, 'decoder': 0.0}
pk_total = 0
for i in range(nb + 5):
    if i == 5:
        t = dict.fromkeys(t, 0.0)
        pk_total = 0
    a = time.perf_counter()
    part = run.feed_resident(blocks.data_ptr() + (i % 16) * esz)
    b = time.perf_counter()
    tail = run.demod.overlapTail(part['rec'])
    c = time.perf_counter()
    d = run.feed_host(part)
    e = time.perf_counter()
    pk, _, _ = dec.findFrames(d['data'], 0)
    f = time.perf_counter()
    pk_total += len(pk)
    t['device'] += b - a
    t['tail'] += c - b
    t['host_stage'] += e - c
    t['decoder'] += f - e
print({k: round(v / nb * 1e3, 4) for k, v in t.items()}, 'ms per block;', pk_total / nb, 'packets per block;', len(d['data']), 'bits per block')
run.close()
