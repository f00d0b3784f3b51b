"""Soak of the streaming receive chain: cycles of (new DemodulatorRunner + Decoder, a stream of blocks through run_stream,
close) and one long stream on a single handle, watching device memory (hipMemGetInfo through torch), host RSS and open
file descriptors.  A leak of streams, events, graphs, pinned buffers or device buffers shows as growth per cycle.

    python tools/soak.py [cycles=40] [blocks_per_cycle=30] [long_blocks=4000]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycusdr_amd import config as cfg, signals as sg              # noqa: E402
from pycusdr_amd.decoder import Decoder                            # noqa: E402
from pycusdr_amd.demodulator_process import DemodulatorRunner      # noqa: E402
from pycusdr_amd.hostcpu import quiet_blas                         # noqa: E402
from pycusdr_amd.protocol import loadProtocol                      # noqa: E402


def rss_mb():
    with open('/proc/self/statm') as f:
        return int(f.read().split()[1]) * os.sysconf('SC_PAGE_SIZE') / 2 ** 20


def nfds():
    return len(os.listdir('/proc/self/fd'))


def main():
    import torch
    quiet_blas()
    cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    long_blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
    bs = 15
    N = 1 << bs
    sig, _ = sg.get_padded_packet('GMSK', 16, 153600)
    rng = np.random.RandomState(7)
    base = np.concatenate([sg.awgn(sig, 8.0, rng=rng).astype(np.complex64) for _ in range(4)])
    step = N - 1024

    def stream(nblocks):
        need = nblocks * step + 1024
        reps = -(-need // len(base))
        s = np.tile(base, reps)[:need]
        return (s[i:i + 16384] for i in range(0, len(s), 16384))

    def snapshot():
        torch.cuda.synchronize()
        free, _ = torch.cuda.mem_get_info()
        return free / 2 ** 20, rss_mb(), nfds()

    protos = ['bench_GMSK', 'bench_FSK', 'bench_BPSK']
    marks = []
    for c in range(cycles):
        pname = protos[c % len(protos)]
        conf = cfg.bench_config(pname, blockSize=bs, doppCarrierSteps=32 + 16 * (c % 3))
        proto = loadProtocol(pname)(conf=conf)
        run = DemodulatorRunner(conf, proto, 'UHF-H')
        dec = Decoder(conf, proto)
        dec.prepare()
        res, pk = run.run_stream(stream(per), decoder=dec)
        run.close()
        dec.close() if hasattr(dec, 'close') else None
        del run, dec
        if c in (2, cycles // 2, cycles - 1):
            marks.append((c,) + snapshot())
            print(f'cycle {c}: device free {marks[-1][1]:.0f} MiB, rss {marks[-1][2]:.0f} MiB, fds {marks[-1][3]}', flush=True)
    # first half: allocator pools and code objects settle; a leak keeps growing through the second half
    d_dev = marks[1][1] - marks[-1][1]
    d_rss = marks[-1][2] - marks[1][2]
    d_fd = marks[-1][3] - marks[1][3]
    n = marks[-1][0] - marks[1][0]
    print(f'per cycle over the second half: device {d_dev / n:+.3f} MiB, rss {d_rss / n:+.3f} MiB, fds {d_fd / n:+.2f}')
    # one long stream on one handle
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=64)
    proto = loadProtocol('bench_GMSK')(conf=conf)
    run = DemodulatorRunner(conf, proto, 'UHF-H')
    dec = Decoder(conf, proto)
    dec.prepare()
    run.run_stream(stream(50), decoder=dec)
    a = snapshot()
    t0 = time.perf_counter()
    res, pk = run.run_stream(stream(long_blocks), decoder=dec)
    dt = time.perf_counter() - t0
    b = snapshot()
    print(f'long stream: {len(res)} blocks, {len(pk)} packets, {len(res) * step / dt / 1e6:.0f} Msamples/s; device {a[0] - b[0]:+.1f} MiB, '
          f'rss {b[1] - a[1]:+.1f} MiB, fds {b[2] - a[2]:+d}')
    run.close()
    ok = abs(d_dev / n) < 0.5 and d_rss / n < 1.0 and d_fd / n < 0.5 and abs(a[0] - b[0]) < 8 and (b[2] - a[2]) == 0
    print('ok' if ok else 'GROWTH')
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
