"""Per-block wall time of the BER bench's receive loop (examples/benchmark/bench_modem.py), to find where a slow row
spends its time.  For each row (a fresh DemodulatorRunner and decoder, ten bench packets + flush, like one SNR of the
bench) it prints the rate, the median block time and every block that took more than 2 ms, with the CPU time this
thread and the whole process used across that block (a host that was descheduled shows wall >> cpu; a host spinning on
a slow GPU shows wall ~ cpu) and the cgroup's throttling counters before and after the run.

    ROWS=12 python tools/ber_rows.py          # BLAS=default: leave numpy's BLAS pool alone (shows the stall);
                                              # NOGC=1: Python's collector off; MFB_NO_GRAPH=1: launches, not graph replays
"""
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycusdr_amd import config as cfg, signals as sg              # noqa: E402
from pycusdr_amd.decoder import Decoder                            # noqa: E402
from pycusdr_amd.demodulator_process import DemodulatorRunner      # noqa: E402
from pycusdr_amd.hostcpu import cpu_share, quiet_blas              # noqa: E402
from pycusdr_amd.protocol import loadProtocol                      # noqa: E402


def cpu_stat():
    for p in ('/sys/fs/cgroup/cpu.stat', '/sys/fs/cgroup/cpu/cpu.stat', '/sys/fs/cgroup/cpu,cpuacct/cpu.stat'):
        try:
            with open(p) as f:
                return {k: int(v) for k, v in (line.split() for line in f)}
        except OSError:
            continue
    return {}


def main():
    if os.environ.get('NOGC'):
        gc.disable()
    if os.environ.get('BLAS') != 'default':
        quiet_blas()
    print('CPU share', cpu_share(), 'of', os.cpu_count(), flush=True)
    bs = 15
    N = 1 << bs
    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=64)
    proto = loadProtocol('bench_GMSK')(conf=conf)
    sig, _ = sg.get_padded_packet('GMSK', 16, 153600)
    before = cpu_stat()
    print('cgroup cpu.stat before:', {k: v for k, v in before.items() if 'throttl' in k or k == 'nr_periods'}, flush=True)
    for row in range(int(os.environ.get('ROWS', '8'))):
        rng = np.random.RandomState(1000 + row)
        run = DemodulatorRunner(conf, proto, 'UHF-H')
        parts = [sg.awgn(sig, 5.0, rng=rng).astype(np.complex64) for _ in range(10)]
        parts.append((1e-3 * (rng.standard_normal(2 * N) + 1j * rng.standard_normal(2 * N))).astype(np.complex64))
        stream = np.concatenate(parts)
        dec = Decoder(conf, proto)
        dec.prepare()
        stamps = []
        t0 = (time.perf_counter(), time.thread_time(), time.process_time())
        run.run_stream((stream[i:i + 16384] for i in range(0, len(stream), 16384)), decoder=dec,
                       sink=lambda d: stamps.append((time.perf_counter(), time.thread_time(), time.process_time())))
        dt = time.perf_counter() - t0[0]
        s = np.array([t0] + stamps)
        d = np.diff(s, axis=0) * 1e3
        big = [f'block {i}: wall {w:.1f} ms, thread cpu {t:.1f} ms, process cpu {p:.1f} ms' for i, (w, t, p) in enumerate(d) if w > 2.0]
        print(f'row {row}: {len(stream) / dt / 1e6:7.1f} Msps, {dt * 1e3:6.1f} ms, {len(d)} blocks, median {np.median(d[:, 0]):.3f} ms; '
              f'slow: {big}', flush=True)
        run.close()
    after = cpu_stat()
    print('cgroup cpu.stat after: ', {k: v for k, v in after.items() if 'throttl' in k or k == 'nr_periods'}, flush=True)


if __name__ == '__main__':
    main()
