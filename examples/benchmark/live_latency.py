"""What a live source pays for batches: a producer thread hands the receive loop 4096-sample chunks at a set pace through a queue
(the reference's ZeroMQ subscriber in miniature, sigFIFO.py:156-163), the loop runs ``run_stream(drain_marked(poll, wait))`` --
up to 32 blocks (2^20 samples) per device call while there is a backlog, every complete block at once while there is none.  Printed per pace: the
mean number of blocks per device call and the latency from a block's last sample to its result dict.  Then the same paces with a PLAIN
iterator over the queue (no markers, nothing configured): the loop decides from the time each chunk took to come (round 6).
usage: python examples/benchmark/live_latency.py [log2N] [blocks per pace]"""
import os
import queue
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pycusdr_amd import config as cfg, signals as sg                 # noqa: E402
from pycusdr_amd.demodulator_process import DemodulatorRunner         # noqa: E402
from pycusdr_amd.hostcpu import quiet_blas                            # noqa: E402
from pycusdr_amd.protocol import loadProtocol                         # noqa: E402

CHUNK = 4096


def one_pace(run, stream, msamples_per_s, marked=True):
    """Feed ``stream`` at ``msamples_per_s`` (None: as fast as the queue takes it); returns (blocks, mean batch, latencies ms)."""
    q = queue.Queue(maxsize=4096)
    period = CHUNK / (msamples_per_s * 1e6) if msamples_per_s else 0.0

    def producer():
        # (sleeps, never spins: a spinning Python thread would hold the interpreter lock against the receive loop; paces faster
        # than a chunk per 0.5 ms hand over the chunks that are due at each wake-up)
        t0 = time.perf_counter()
        for k, i in enumerate(range(0, len(stream), CHUNK)):
            if period:
                due = t0 + (k + 1) * period
                now = time.perf_counter()
                if due - now > 2e-4:
                    time.sleep(due - now)
            q.put(stream[i:i + CHUNK])
        q.put(None)
    th = threading.Thread(target=producer, daemon=True)

    def poll():
        try:
            c = q.get_nowait()
        except queue.Empty:
            return None
        if c is None:
            q.put(None)              # the end marker stays for wait()
        return c

    def wait():
        return q.get()
    sizes = []
    inner = run.demod.beginBlocks
    run.demod.beginBlocks = lambda which, nb, **kw: (sizes.append(nb), inner(which, nb, **kw))[1]
    lat = []
    th.start()
    try:
        def plain():                 # a source that simply blocks until its next chunk exists
            while True:
                c = q.get()
                if c is None:
                    return
                yield c
        run.run_stream(DemodulatorRunner.drain_marked(poll, wait) if marked else plain(), sink=lambda d: lat.append(d['latency_ms']))
    finally:
        run.demod.beginBlocks = inner
        th.join()
    return len(lat), (sum(sizes) / len(sizes) if sizes else 0.0), np.asarray(lat)


def main():
    quiet_blas()
    log2N = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    nblocks = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    N, ov = 1 << log2N, 1 << 10
    conf = cfg.bench_config('bench_GMSK', blockSize=log2N, doppCarrierSteps=64)
    p = loadProtocol('bench_GMSK')(conf=conf)
    stream = sg.s1_stream(nblocks, N, ov, 'GMSK', snr_db=12.0, seed=3)[ov:]
    stream.flags.writeable = False
    run = DemodulatorRunner(conf, p, 'UHF-H')
    try:
        one_pace(run, stream[:20 * (N - ov)], None)          # warm-up: library, graphs of several batch sizes
        print(f'N = 2^{log2N}, 64 bins, {nblocks} blocks per pace; block = {(N - ov) / 1e3:.1f} ksamples')
        print('pace (Msamples/s) | blocks per device call | latency ms: median / 90 % / max | block period ms')
        for marked in (True, False):
            print('-- the source marks where it would block (drain_marked)' if marked else '-- a plain iterator, nothing configured (adaptive)')
            for pace in (2, 20, 100, 300, 600, None):
                n, mean_b, lat = one_pace(run, stream, pace, marked=marked)
                period = (N - ov) / (pace * 1e6) * 1e3 if pace else 0.0
                print(f'{pace if pace else "unpaced":>8} | {mean_b:5.2f} | {np.median(lat):7.3f} / {np.percentile(lat, 90):7.3f} / {lat.max():7.3f} | '
                      f'{period:6.3f}   ({n} blocks)', flush=True)
    finally:
        run.close()


if __name__ == '__main__':
    main()
