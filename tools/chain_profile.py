"""cProfile of the whole receive chain (chunks -> Demodulator -> Decoder) at a given block size: where the host
time goes around the device calls.  usage: chain_profile.py [log2N] [nRuns] [modulation] [doppler bins] [blocks per call] [decode 0|1]"""
import cProfile
import importlib.util
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('bench_modem', os.path.join(ROOT, 'examples', 'benchmark', 'bench_modem.py'))
bm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bm)
bm.quiet_blas()
log2N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
mod = sys.argv[3] if len(sys.argv) > 3 else 'GMSK'
D = int(sys.argv[4]) if len(sys.argv) > 4 else 256
B = int(sys.argv[5]) if len(sys.argv) > 5 else 1
decode = bool(int(sys.argv[6])) if len(sys.argv) > 6 else True
bm.run_snr(mod, 2, 12.0, log2N, 'transforms', 1, D, blocks_per_call=B, decode=decode)        # warm-up (library load, allocations)
pr = cProfile.Profile()
orig = bm.DemodulatorRunner.run_stream


def profiled(self, *a, **k):        # the stream alone: the stimulus generator is not part of the chain
    pr.enable()
    try:
        return orig(self, *a, **k)
    finally:
        pr.disable()


bm.DemodulatorRunner.run_stream = profiled
r = bm.run_snr(mod, n, 12.0, log2N, 'transforms', 2, D, blocks_per_call=B, decode=decode)
print({k: v for k, v in r.items() if k != 'bitErrors'})
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
