"""Whole-chain rate (ring buffer -> H2D -> Doppler search -> demodulation -> decoder -> packets) at a block size and bin
count, sequential and pipelined.  usage: chain_rate.py [log2N] [nRuns] [doppler bins] [modulation]"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('bench_modem', os.path.join(ROOT, 'examples', 'benchmark', 'bench_modem.py'))
bm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bm)
bm.quiet_blas()          # numpy's BLAS workers must not spend the container's CPU quota (pycusdr_amd/hostcpu.py)
log2N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
D = int(sys.argv[3]) if len(sys.argv) > 3 else 256
mod = sys.argv[4] if len(sys.argv) > 4 else 'GMSK'
bm.run_snr(mod, 2, 12.0, log2N, 'transforms', 1, D)
for search in ('transforms', 'energy'):
    for pipelined in (False, True):
        r = bm.run_snr(mod, n, 12.0, log2N, search, 2, D, pipelined)
        print(f"{mod} N=2^{log2N} D={D} search={search:10s} pipelined={pipelined!s:5s}: {r['ksamples_per_s'] / 1e3:8.1f} Msamples/s, "
              f"{r['blocks']} blocks, packets {r['packets']}/{r['sent']}, BER {r['BER']:.2e}", flush=True)
