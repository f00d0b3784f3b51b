"""The device side of the ONE-block receive path alone (mfb_receive_block_begin / _end): blocks begun back to back from the two
page-locked input buffers, each collected one block behind, no host stages -- forward transform, search, pick, matched filters at the
picked shift, envelope transform, rate, centres, one read-back, the 8 MiB host-to-device copy beside it.  One stream against two
(`MFBank.set_batch_overlap`: the next block's search beside this block's demodulation stage).
usage: python3 tools/block_device_rate.py [log2N] [bins] [blocks] [overlap 0|1] [protocol]"""
import sys
import time
import numpy as np
sys.path.insert(0, '.')
from pycusdr_amd.hostcpu import quiet_blas  # noqa: E402
quiet_blas()
from pycusdr_amd import config as cfg, signals as sg  # noqa: E402
from pycusdr_amd.demodulator_process import DemodulatorRunner  # noqa: E402
from pycusdr_amd.protocol import loadProtocol  # noqa: E402

log2N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
nblk = int(sys.argv[3]) if len(sys.argv) > 3 else 200
overlap = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
pname = sys.argv[5] if len(sys.argv) > 5 else 'bench_GMSK'
N, ov = 1 << log2N, 1 << 10
conf = cfg.cc11xx_config(blockSize=log2N, doppCarrierSteps=D) if pname == 'CC11xx' else cfg.bench_config(pname, blockSize=log2N, doppCarrierSteps=D)
p = loadProtocol(pname)(conf=conf)
run = DemodulatorRunner(conf, p, 'UHF-H')
sig = sg.s1_stream(2, N, ov, 'GMSK', snr_db=12.0, seed=3)
bufs = (run.demod.bank.input, run.demod.bank.input2)
bufs[0][:] = sig[:N]
bufs[1][:] = sig[N - ov:2 * N - ov]
run.demod.bank.set_batch_overlap(overlap)
names = ('pinned', 'pinned2')


def loop(n):
    run.demod.beginBlock(0, source=names[0])
    for k in range(1, n):
        run.demod.beginBlock(k & 1, source=names[k & 1])
        run.demod.endBlock((k - 1) & 1)
    run.demod.endBlock((n - 1) & 1)


loop(40)
for rep in range(3):
    t0 = time.perf_counter()
    loop(nblk)
    dt = (time.perf_counter() - t0) / nblk
    print(f'{pname} N=2^{log2N} D={D} overlap={int(overlap)}: {dt * 1e3:8.4f} ms per block, {(N - ov) / dt / 1e6:8.1f} Msamples/s', flush=True)
run.close()
