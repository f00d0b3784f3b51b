"""cProfile of the whole receive chain (ring buffer -> Demodulator -> Decoder) at a given block size: where the host
time goes around the device calls.  usage: chain_profile.py [log2N] [nRuns] [modulation] [doppler bins]"""
import cProfile
import importlib.util
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('bench_modem', os.path.join(ROOT, 'examples', 'benchmark', 'bench_modem.py'))
bm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bm)
log2N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
mod = sys.argv[3] if len(sys.argv) > 3 else 'GMSK'
D = int(sys.argv[4]) if len(sys.argv) > 4 else 256
bm.run_snr(mod, 2, 12.0, log2N, 'transforms', 1, D)        # warm-up (library load, allocations)
pr = cProfile.Profile()
pr.enable()
r = bm.run_snr(mod, n, 12.0, log2N, 'transforms', 2, D)
pr.disable()
print({k: v for k, v in r.items() if k != 'bitErrors'})
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
