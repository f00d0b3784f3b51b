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


def soak(cycles=40, per=30, long_blocks=4000, budget_s=None, min_cycles=12, log=print):
    """Run the soak; returns a dict of the measured growth figures and 'ok'.  ``budget_s``: stop the create / stream / close
    cycles early (never before ``min_cycles``) once that many seconds have passed."""
    import torch
    quiet_blas()
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
    t_start = time.perf_counter()
    done = 0
    for c in range(cycles):
        pname = protos[c % len(protos)]
        conf = cfg.bench_config(pname, blockSize=bs, doppCarrierSteps=32 + 16 * (c % 3))
        if c % 4 >= 2:              # every other pair of cycles: several blocks per device call, the stream stages on the device
            conf['GPU']['UHF'].setdefault('HIP', {})['blocks_per_call'] = 3 + c % 5
        proto = loadProtocol(pname)(conf=conf)
        run = DemodulatorRunner(conf, proto, 'UHF-H')
        dec = Decoder(conf, proto)
        dec.prepare()
        res, pk = run.run_stream(stream(per), decoder=dec)
        if c % 2:                   # the same runner again (it goes on in whichever page-locked buffer the first call ended in)
            run.run_stream(stream(3), decoder=dec)
        run.close()
        dec.close() if hasattr(dec, 'close') else None
        del run, dec
        done = c + 1
        marks.append((c,) + snapshot())
        if budget_s is not None and done >= min_cycles and time.perf_counter() - t_start > budget_s:
            break
    # first half: allocator pools and code objects settle; a leak keeps growing through the second half
    mid, last = marks[len(marks) // 2], marks[-1]
    for m in (marks[min(2, len(marks) - 1)], mid, last):
        log(f'cycle {m[0]}: device free {m[1]:.0f} MiB, rss {m[2]:.0f} MiB, fds {m[3]}')
    n = max(last[0] - mid[0], 1)
    d_dev, d_rss, d_fd = (mid[1] - last[1]) / n, (last[2] - mid[2]) / n, (last[3] - mid[3]) / n
    log(f'{done} cycles in {time.perf_counter() - t_start:.1f} s; per cycle over the second half: device {d_dev:+.3f} MiB, rss {d_rss:+.3f} MiB, fds {d_fd:+.2f}')
    # one long stream on one handle
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=64)
    proto = loadProtocol('bench_GMSK')(conf=conf)
    run = DemodulatorRunner(conf, proto, 'UHF-H')
    dec = Decoder(conf, proto)
    dec.prepare()
    run.run_stream(stream(51), decoder=dec)          # an odd number of blocks: the long stream starts in the second buffer
    a = snapshot()
    t0 = time.perf_counter()
    res, pk = run.run_stream(stream(long_blocks), decoder=dec)
    dt = time.perf_counter() - t0
    b = snapshot()
    log(f'long stream: {len(res)} blocks, {len(pk)} packets, {len(res) * step / dt / 1e6:.0f} Msamples/s; device {a[0] - b[0]:+.1f} MiB, '
        f'rss {b[1] - a[1]:+.1f} MiB, fds {b[2] - a[2]:+d}')
    run.close()
    # ... and the same with 16 blocks per device call (windows, batch buffers, block graphs per window / slot / parity, carries)
    confB = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=64)
    confB['GPU']['UHF'].setdefault('HIP', {})['blocks_per_call'] = 16
    runB = DemodulatorRunner(confB, proto, 'UHF-H')
    decB = Decoder(confB, proto)
    decB.prepare()
    runB.run_stream(stream(51), decoder=decB)
    aB = snapshot()
    t0 = time.perf_counter()
    resB, pkB = runB.run_stream(stream(long_blocks), decoder=decB)
    dtB = time.perf_counter() - t0
    bB = snapshot()
    log(f'long stream, 16 blocks per call: {len(resB)} blocks, {len(pkB)} packets, {len(resB) * step / dtB / 1e6:.0f} Msamples/s; '
        f'device {aB[0] - bB[0]:+.1f} MiB, rss {bB[1] - aB[1]:+.1f} MiB, fds {bB[2] - aB[2]:+d}; stage blocks {getattr(runB.demod, "stage_blocks", 0)}')
    runB.close()
    same = len(resB) == len(res) and len(pkB) == len(pk) and all(np.array_equal(x['data'], y['data']) for x, y in zip(res[::97], resB[::97]))
    out = {'batched_blocks': len(resB), 'batched_packets': len(pkB), 'batched_device_mib': aB[0] - bB[0], 'batched_fds': bB[2] - aB[2],
           'batched_rss_mib': bB[1] - aB[1], 'batched_equals_one_block_loop': bool(same), 'batched_msamples': len(resB) * step / dtB / 1e6,
           'cycles': done, 'device_mib_second_half': mid[1] - last[1], 'rss_mib_second_half': last[2] - mid[2], 'device_mib_per_cycle': d_dev, 'rss_mib_per_cycle': d_rss, 'fds_per_cycle': d_fd,
           'long_blocks': len(res), 'long_packets': len(pk), 'long_device_mib': a[0] - b[0], 'long_rss_mib': b[1] - a[1],
           'long_fds': b[2] - a[2], 'long_msamples': len(res) * step / dt / 1e6}
    # (device figures one-sided: the allocator handing a granule back during a stream is not growth)
    out['ok'] = bool(abs(d_dev) < 0.5 and d_rss < 1.0 and d_fd < 0.5 and -64 < out['long_device_mib'] < 8.5 and out['long_fds'] == 0
                     and -64 < out['batched_device_mib'] < 8.5 and out['batched_fds'] == 0 and same)
    return out


def main():
    cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    per = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    long_blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
    out = soak(cycles, per, long_blocks, log=lambda m: print(m, flush=True))
    print('ok' if out['ok'] else 'GROWTH')
    return 0 if out['ok'] else 1


if __name__ == '__main__':
    sys.exit(main())
