"""Whole-chain rate (chunks -> page-locked buffer -> H2D -> Doppler search -> demodulation -> host stages [-> decoder -> packets])
at a block size and bin count: the one-block loop, the loop with B blocks per device call, with and without the decoder (in the
reference the decoder is another process: the receive loop proper, DP:284-338, ends at the send).
usage: chain_rate.py [log2N] [nRuns] [doppler bins] [modulation] [B,B,...]"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('bench_modem', os.path.join(ROOT, 'examples', 'benchmark', 'bench_modem.py'))
bm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bm)
bm.quiet_blas()          # numpy's BLAS workers must not spend the container's CPU quota (pycusdr_amd/hostcpu.py)
log2N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 240
D = int(sys.argv[3]) if len(sys.argv) > 3 else 256
mod = sys.argv[4] if len(sys.argv) > 4 else 'GMSK'
Bs = [x if x == 'auto' else int(x) for x in sys.argv[5].split(',')] if len(sys.argv) > 5 else [1, 4, 8, 16]      # 'auto': nothing configured
if mod == 'CC11xx':
    # the reference's production protocol at its own geometry (config/CC11xx.json: 128 samples per symbol, 384-tap filters, IF offset)
    stim = bm.make_cc11xx_stream(n, 12.0, log2N, 3)
    bm.run_cc11xx(n, 12.0, log2N, D, stimulus=bm.make_cc11xx_stream(4, 12.0, log2N, 3))
    for B in Bs:
        for decode in (True, False):
            r = max((bm.run_cc11xx(n, 12.0, log2N, D, blocks_per_call=B, decode=decode, stimulus=stim) for _ in range(3)),
                    key=lambda q: q['ksamples_per_s'])
            print(f"CC11xx N=2^{log2N} D={D} blocks_per_call={B!s:>4s} decoder={decode!s:5s}: {r['ksamples_per_s'] / 1e3:8.1f} Msamples/s, "
                  f"{r['blocks']} blocks, frames {r['frames']}/{r['sent']} ({r['packets']} candidates)", flush=True)
    sys.exit(0)
bm.run_snr(mod, 2, 12.0, log2N, 'transforms', 1, D)
stim = bm.make_stream(mod, n, 12.0, log2N, 2)
for B in Bs:
    for decode in (True, False):
        # host-bound loop on shared CPUs: the best of three runs on the same stimulus
        r = max((bm.run_snr(mod, n, 12.0, log2N, 'transforms', 2, D, False, blocks_per_call=B, decode=decode, stimulus=stim) for _ in range(3)),
                key=lambda q: q['ksamples_per_s'])
        print(f"{mod} N=2^{log2N} D={D} blocks_per_call={B!s:>4s} decoder={decode!s:5s}: {r['ksamples_per_s'] / 1e3:8.1f} Msamples/s, "
              f"{r['blocks']} blocks, packets {r['packets']}/{r['sent']}, BER {r['BER']:.2e}", flush=True)
if len(sys.argv) > 6 and sys.argv[6] == 'all':
    for search in ('energy',):
        for pipelined in (False, True):
            r = bm.run_snr(mod, n, 12.0, log2N, search, 2, D, pipelined)
            print(f"{mod} N=2^{log2N} D={D} search={search:10s} pipelined={pipelined!s:5s}: {r['ksamples_per_s'] / 1e3:8.1f} Msamples/s, "
                  f"{r['blocks']} blocks, packets {r['packets']}/{r['sent']}, BER {r['BER']:.2e}", flush=True)
