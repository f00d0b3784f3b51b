"""Write profiles/<round>_pmc_twopass.md and the two-pass entry of profiles/pmc_traffic.json from `tools/prof_bench.sh <tag>_twopass --path twopass`.
usage: python tools/pmc_twopass.py r5 r05"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def newest(pattern):
    best = {}
    for f in glob.glob(pattern):
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())


tag, rnd = sys.argv[1], sys.argv[2]
t = tag + '_twopass'
ks = newest(f'gpurun_out/{t}/trace/*/*kernel_stats.csv')[0]
shutil.copy(ks, f'profiles/{rnd}_bench_twopass_kernel_stats.csv')
stats = {r['Name']: r for r in csv.DictReader(open(ks))}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in newest(f'gpurun_out/{t}/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
N, M, Dc = 1 << 20, 8, 128
rows = []
for name, alg, what in (('void k_pass1<256, 0>(P1Args)', 8.0 * Dc * M * N + 8.0 * N * (1 + M), 'shift-multiply, column FFT, twiddle -> Z'),
                        ('void k_pass2<4096, 0>(P2Args)', 8.0 * Dc * M * N + 4.0 * Dc * M * 16, 'row FFT, |.|^2 sums')):
    st, m = stats[name], agg[name]
    f, w = m['FETCH_SIZE'], m['WRITE_SIZE']
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    b = 2 * fm * 1024 + wm * 1024
    avg = float(st['AverageNs']) / 1e3
    rows.append((name, what, avg, int(st['Calls']), alg, alg / (avg * 1e-6) / 1e12, fm, min(f), max(f), wm, b, b / alg, len(f)))
with open(f'profiles/{rnd}_pmc_twopass.md', 'w') as o:
    o.write(f"# Round {int(rnd[1:])} -- the two-pass fallback at HEAD: kernel trace and HBM-traffic counters re-cut\n\n"
            f"`tools/prof_bench.sh {t} --path twopass --steps 12` on the box of `{rnd}_bench.json` (C2: D = 256, M = 8, N = 2^20; 128 bins per launch,\n"
            f"two launches of each pass per block); written by `tools/pmc_twopass.py`.  Nothing of this path changed in round {int(rnd[1:])} (round 5 gave the plain forward\n"
            f"transforms a row index for batches of blocks, `k_pass1<FWDC/FWDR>`: same arithmetic); the figures confirm the earlier rounds'.\n\n"
            "| kernel | average per launch (kernel trace) | algorithmic bytes per launch | achieved | of 8 TB/s | FETCH_SIZE (KiB, mean; min ... max) | WRITE_SIZE (KiB) | counter bytes per launch | over algorithmic |\n|---|---|---|---|---|---|---|---|---|\n")
    for (name, what, avg, calls, alg, tbs, fm, fmin, fmax, wm, b, ratio, n) in rows:
        short = name.replace('void ', '').replace('(P1Args)', '').replace('(P2Args)', '')
        o.write(f"| `{short}` ({what}) | {avg:.1f} us ({calls} launches) | {alg / 1e9:.3f} GB | {tbs:.2f} TB/s | {tbs / 8:.2f} | "
                f"{fm:,.0f} ({fmin:,.0f} ... {fmax:,.0f}; {n} dispatches) | {wm:,.0f} | **{b / 1e9:.3f} GB** | x {ratio:.3f} |\n")
    o.write("\nbytes = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; separate `--pmc` passes).\n"
            "Pass 2 moves exactly the intermediate; pass 1 writes exactly the intermediate and fetches the 64 MiB filter bank again per Doppler stream --\n"
            "fabric traffic served by the 256 MiB Infinity Cache (it varies between dispatches with how much of the bank the XCD L2s still hold).\n"
            "`profiles/pmc_traffic.json` carries the pass-1 / pass-2 bytes into the line's `roofline_other_path.traffic`.\n")
d = json.load(open('profiles/pmc_traffic.json'))
d['D256_M8_N20_chunk128'] = {'pass1': int(rows[0][10]), 'pass2': int(rows[1][10]),
                            'source': f'profiles/{rnd}_pmc_twopass.md: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of round {int(rnd[1:])} over bench.py --path twopass; '
                                      'stored from the profile, not measured by the bench run'}
json.dump(d, open('profiles/pmc_traffic.json', 'w'), indent=1)
print(open(f'profiles/{rnd}_pmc_twopass.md').read())
