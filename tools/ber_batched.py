"""BER rows of the reference's bench (examples/benchmark/bench_modem.py) with one block per device call and with B blocks per call:
and with nothing configured (the adaptive loop): the rows must be identical -- packets found, bit errors per packet -- for every modulation.
usage: ber_batched.py [nRuns] [B]"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('bench_modem', os.path.join(ROOT, 'examples', 'benchmark', 'bench_modem.py'))
bm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bm)
bm.quiet_blas()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ok = True
print('| modulation | SNR | packets (B = 1) | packets (B = %d) | packets (auto) | BER (B = 1) | BER (B = %d) | bit errors equal | Msamples/s B = 1 -> B = %d -> auto |' % (B, B, B))
print('|---|---|---|---|---|---|---|---|---|')
for mod in ('GMSK', 'FSK', 'GFSK', 'BPSK'):
    for k, snr in enumerate((2.0, 6.0, 10.0)):
        stim = bm.make_stream(mod, n, snr, 15, 1000 + k)
        a = bm.run_snr(mod, n, snr, 15, 'transforms', 1000 + k, 64, stimulus=stim)
        b = bm.run_snr(mod, n, snr, 15, 'transforms', 1000 + k, 64, blocks_per_call=B, stimulus=stim)
        c = bm.run_snr(mod, n, snr, 15, 'transforms', 1000 + k, 64, blocks_per_call='auto', stimulus=stim)
        same = a['bitErrors'] == b['bitErrors'] == c['bitErrors'] and a['packets'] == b['packets'] == c['packets']
        ok = ok and same
        print(f"| {mod} | {snr:.0f} dB | {a['packets']}/{a['sent']} | {b['packets']}/{b['sent']} | {c['packets']}/{c['sent']} | {a['BER']:.3e} | {b['BER']:.3e} | {same} | "
              f"{a['ksamples_per_s'] / 1e3:.0f} -> {b['ksamples_per_s'] / 1e3:.0f} -> {c['ksamples_per_s'] / 1e3:.0f} |", flush=True)
print('identical' if ok else 'DIFFERENT')
sys.exit(0 if ok else 1)
