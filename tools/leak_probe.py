"""Where does a create / use / close cycle leave memory behind?  Each stage runs `cycles` times; device free memory (hipMemGetInfo),
host RSS and open file descriptors are read before and after (after a few settling cycles).  Stages go from the bare runtime (streams,
events, page-locked and device buffers through ctypes on libamdhip64) over a bare handle to the runner of tools/soak.py.

    python tools/leak_probe.py [cycles=200] [stage,stage,...]
"""
import ctypes
import gc
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rss_mb():
    with open('/proc/self/statm') as f:
        return int(f.read().split()[1]) * os.sysconf('SC_PAGE_SIZE') / 2 ** 20


def main():
    import torch
    from pycusdr_amd import config as cfg, signals as sg
    from pycusdr_amd.decoder import Decoder
    from pycusdr_amd.demodulator import UHF
    from pycusdr_amd.demodulator_process import DemodulatorRunner
    from pycusdr_amd.hostcpu import quiet_blas
    from pycusdr_amd.protocol import loadProtocol
    quiet_blas()
    cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    only = sys.argv[2].split(',') if len(sys.argv) > 2 else None
    torch.zeros(1, device='cuda')
    hip = ctypes.CDLL('libamdhip64.so')
    bs = 15
    N = 1 << bs
    sig, _ = sg.get_padded_packet('GMSK', 16, 153600)
    base = np.concatenate([sg.awgn(sig, 8.0, rng=np.random.RandomState(7)).astype(np.complex64) for _ in range(4)])
    step = N - 1024

    def stream(nblocks):
        need = nblocks * step + 1024
        s = np.tile(base, -(-need // len(base)))[:need]
        return (s[i:i + 16384] for i in range(0, len(s), 16384))

    def snap():
        torch.cuda.synchronize()
        free, _ = torch.cuda.mem_get_info()
        return free / 2 ** 20, rss_mb(), len(os.listdir('/proc/self/fd'))

    def hip_streams():
        s = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
        assert hip.hipStreamDestroy(s) == 0

    def hip_priority_streams():
        s = ctypes.c_void_p()
        assert hip.hipStreamCreateWithPriority(ctypes.byref(s), 1, -1) == 0
        assert hip.hipStreamDestroy(s) == 0

    def hip_events():
        for _ in range(16):
            e = ctypes.c_void_p()
            assert hip.hipEventCreateWithFlags(ctypes.byref(e), 2) == 0
            assert hip.hipEventDestroy(e) == 0

    def hip_pinned():
        p = ctypes.c_void_p()
        assert hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 20), 0) == 0
        assert hip.hipHostFree(p) == 0

    def hip_device():
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(3 << 20)) == 0
        assert hip.hipFree(p) == 0

    conf = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=32)
    proto = loadProtocol('bench_GMSK')(conf=conf)

    def handle_only():
        d = UHF.Demodulator(conf, proto, 'UHF-H')
        d.close()

    x0 = np.tile(base, 2)[:N].copy()

    def handle_search():
        d = UHF.Demodulator(conf, proto, 'UHF-H')
        d.bank.upload(x0)
        d.bank.find_carrier()
        d.close()

    def handle_blocks():
        d = UHF.Demodulator(conf, proto, 'UHF-H')
        for _ in range(4):          # the recorded graph of a block appears with its second use
            d.uploadAndFindCarrier(x0.copy())
            d.demodulate()
        d.close()

    def runner(bpc, decode=True, collect=False, blocks=12):
        def f():
            c2 = cfg.bench_config('bench_GMSK', blockSize=bs, doppCarrierSteps=32)
            if bpc:
                c2['GPU']['UHF'].setdefault('HIP', {})['blocks_per_call'] = bpc
            run = DemodulatorRunner(c2, proto, 'UHF-H')
            dec = None
            if decode:
                dec = Decoder(c2, proto)
                dec.prepare()
            run.run_stream(stream(blocks), decoder=dec)
            run.close()
            del run, dec
            if collect:
                gc.collect()
        return f

    def decoder_only():
        dec = Decoder(conf, proto)
        dec.prepare()
        del dec

    def live():
        import collections
        names = ('MFBank', 'SyncFinder', 'Decoder', 'DemodulatorRunner', 'HostCopy', 'Demodulator')
        cnt = collections.Counter(type(o).__name__ for o in gc.get_objects() if type(o).__name__ in names)
        with open('/proc/self/status') as fh:
            thr = [ln.split()[1] for ln in fh if ln.startswith('Threads:')][0]
        return ', '.join(f'{k} {v}' for k, v in sorted(cnt.items())) + f'; threads {thr}'

    def handle_windows(nb):
        def f():
            d = UHF.Demodulator(conf, proto, 'UHF-H')
            d.bank.windows(nb, step)
            d.close()
        return f

    def batch_handle(stages_on, passes):
        def f():
            d = UHF.Demodulator(conf, proto, 'UHF-H')
            w = d.blockWindows(16)
            if stages_on:
                d.enableStreamStages()
                d.seedStreamStages()
            need = 12 * step + 1024
            w[0][:need] = np.tile(base, -(-need // len(base)))[:need]
            for _ in range(passes):
                d.beginBlocks(0, 12, source='window')
                d.waitBlocks(0)
            d.close()
        return f

    def host_copy():
        from pycusdr_amd.mfbank import HostCopy
        hc = HostCopy()
        dst = np.zeros(1 << 16, np.complex64)
        hc.submit(dst, 0, base[:4096])
        hc.drain()
        hc.close()

    def pinned_big():
        p = ctypes.c_void_p()
        assert hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(9 << 20), 0) == 0
        assert hip.hipHostFree(p) == 0

    def device_big():
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(40 << 20)) == 0
        assert hip.hipFree(p) == 0

    stages = [('hip_streams', hip_streams), ('hip_priority_streams', hip_priority_streams), ('hip_events', hip_events),
              ('hip_pinned', hip_pinned), ('hip_device', hip_device), ('handle_only', handle_only), ('handle_search', handle_search),
              ('handle_blocks', handle_blocks), ('runner_b1', runner(1)), ('runner_b4', runner(4)), ('runner_auto', runner(0)),
              ('runner_b16', runner(16)), ('runner_b16_nodec', runner(16, decode=False)),
              ('runner_b16_gc', runner(16, collect=True)), ('runner_b16_48', runner(16, blocks=48)), ('decoder_only', decoder_only), ('runner_b32', runner(32)), ('handle_windows4', handle_windows(4)),
              ('handle_windows32', handle_windows(32)), ('batch_once', batch_handle(False, 1)), ('batch_x3', batch_handle(False, 3)),
              ('batch_stages_once', batch_handle(True, 1)), ('batch_stages_x3', batch_handle(True, 3)), ('host_copy', host_copy),
              ('pinned_big', pinned_big), ('device_big', device_big)]
    for name, f in stages:
        if only and name not in only:
            continue
        for _ in range(8):
            f()
        a = snap()
        for i in range(cycles):
            f()
            if cycles >= 1000 and (i + 1) % (cycles // 10) == 0:       # a long run: does it level off?
                m = snap()
                print(f'  {name} after {i + 1}: device {a[0] - m[0]:+.1f} MiB, rss {m[1] - a[1]:+.1f} MiB', flush=True)
        b = snap()
        print(f'  live: {live()}')
        print(f'{name:22s} per cycle: device {(a[0] - b[0]) / cycles:+.4f} MiB, rss {(b[1] - a[1]) / cycles:+.4f} MiB, fds {(b[2] - a[2]) / cycles:+.3f}',
              flush=True)


if __name__ == '__main__':
    main()
