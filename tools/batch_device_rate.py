"""The DEVICE side of the batched receive path alone: batches begun back to back from two pre-filled page-locked windows, records
collected one batch behind, no host stages, no decoder -- what mfb_receive_blocks_begin/_end sustain when the host is not the
bound (the receive loop in Python is: ~20 us of host work per block).  The A/B of the batch's two streams
(MFB_BATCH_SPLIT=0/1, profiles/r06_chain.md).
usage: python3 tools/batch_device_rate.py [log2N] [bins] [B] [batches] [stages 0|1] [overlap 0|1]"""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
from pycusdr_amd.hostcpu import quiet_blas  # noqa: E402
quiet_blas()
from pycusdr_amd import config as cfg, signals as sg  # noqa: E402
from pycusdr_amd.decoder import Decoder  # noqa: E402
from pycusdr_amd.demodulator_process import DemodulatorRunner  # noqa: E402
from pycusdr_amd.protocol import loadProtocol  # noqa: E402

log2N = int(sys.argv[1]) if len(sys.argv) > 1 else 15
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
nbat = int(sys.argv[4]) if len(sys.argv) > 4 else 200
stages = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
overlap = bool(int(sys.argv[6])) if len(sys.argv) > 6 else False
N, ov = 1 << log2N, 1 << 10
step = N - ov
conf = cfg.bench_config('bench_GMSK', blockSize=log2N, doppCarrierSteps=D)
p = loadProtocol('bench_GMSK')(conf=conf)
run = DemodulatorRunner(conf, p, 'UHF-H')
dec = Decoder(conf, p)
sig = sg.s1_stream(2 * B, N, ov, 'GMSK', snr_db=12.0, seed=3)
run.demod.bank.set_batch_overlap(overlap)          # (MFB_BATCH_SPLIT in the environment overrides it)
wins = run.demod.blockWindows(B)
wins[0][:] = sig[:B * step + ov]
wins[1][:] = sig[B * step:2 * B * step + ov]
if stages and run.demod.enableStreamStages(dec):
    run.demod.seedStreamStages()
names = ('window', 'window2')


def loop(n):
    run.demod.beginBlocks(0, B, source=names[0])
    for k in range(1, n):
        run.demod.beginBlocks(k & 1, B, source=names[k & 1])
        run.demod.waitBlocks((k - 1) & 1)
    run.demod.waitBlocks((n - 1) & 1)


loop(8)              # graphs recorded, clock settled
best = None
for rep in range(3):
    t0 = time.perf_counter()
    loop(nbat)
    dt = (time.perf_counter() - t0) / nbat
    best = dt if best is None else min(best, dt)
    print(f'N=2^{log2N} D={D} B={B} stages={int(stages)} overlap={int(overlap)}: {dt * 1e6:8.1f} us per batch, {dt / B * 1e6:6.2f} us per block, '
          f'{B * step / dt / 1e6:8.1f} Msamples/s', flush=True)
run.close()
