"""Per-block wall time of the streaming loop on a FRESH handle: shows what the first blocks of a new Demodulator pay
(lazy allocations, code-object loads) against the steady state.  usage: chain_blocks.py [log2N] [bins] [handles]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycusdr_amd import config as cfg, signals as sg            # noqa: E402
from pycusdr_amd.decoder import Decoder                          # noqa: E402
from pycusdr_amd.demodulator_process import DemodulatorRunner    # noqa: E402
from pycusdr_amd.hostcpu import quiet_blas                       # noqa: E402
from pycusdr_amd.protocol import loadProtocol                    # noqa: E402

quiet_blas()

log2N = int(sys.argv[1]) if len(sys.argv) > 1 else 15
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
handles = int(sys.argv[3]) if len(sys.argv) > 3 else 3
N, ov = 1 << log2N, 1 << 10
conf = cfg.bench_config('bench_GMSK', blockSize=log2N, doppCarrierSteps=D)
proto = loadProtocol('bench_GMSK')(conf=conf)
nb = 60
sig = sg.s1_stream(nb, N, ov, 'GMSK', snr_db=12.0, seed=4)
for h in range(handles):
    t_new = time.perf_counter()
    run = DemodulatorRunner(conf, proto, 'UHF-H')
    dec = Decoder(conf, proto)
    t_new = time.perf_counter() - t_new
    stamps = []

    def sink(d):
        stamps.append(time.perf_counter())
    t0 = time.perf_counter()
    run.run_stream((sig[i:i + 16384] for i in range(ov, ov + nb * (N - ov), 16384)), sink=sink, decoder=dec)
    total = time.perf_counter() - t0
    dts = np.diff([t0] + stamps) * 1e3
    run.close()
    print(f'handle {h}: construction {t_new * 1e3:.1f} ms; blocks {len(dts)}; first {dts[0]:.2f} ms, second {dts[1]:.2f}, third {dts[2]:.2f}, '
          f'median {np.median(dts):.3f}, max of the rest {dts[3:].max():.3f}; whole stream {total * 1e3:.1f} ms '
          f'= {len(dts) * (N - ov) / total / 1e6:.1f} Msamples/s (steady state {(N - ov) / np.median(dts) / 1e3:.1f})', flush=True)
