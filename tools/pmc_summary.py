"""Write profiles/<round>_pmc_summary.md, the kernel-stats CSVs and the segment entries of profiles/pmc_traffic.json from the three
tools/prof_bench.sh runs of a round (tags <t>, <t>_cc, <t>_bpsk under gpurun_out/) and the bench line gpurun_out/<t>_bench.json.
usage: python tools/pmc_summary.py r4 r04"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def newest(pattern):
    """gpurun_out/ accumulates the files of earlier runs of the same tag (rocprofv3 prefixes them with its pid): per
    directory only the newest file counts."""
    best = {}
    for f in glob.glob(pattern):
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())

tag, rnd = sys.argv[1], sys.argv[2]
line = json.loads(open(f'gpurun_out/{tag}_bench.json').read().strip().splitlines()[-1])
shutil.copy(f'gpurun_out/{tag}_bench.json', f'profiles/{rnd}_bench.json')
banks = {b['protocol']: b for b in line['config']['other_banks']}
cols = [('', 'C2: GMSK, 48 taps, M = 8', line['roofline'], line['value'], 'kernel_stats'),
        ('_cc', 'CC11xx FSK-2, 384 taps, M = 8', banks['CC11xx']['roofline'], banks['CC11xx']['msamples'], 'cc11xx_kernel_stats'),
        ('_bpsk', 'BPSK, 80 taps, M = 32 (16 unique)', banks['bench_BPSK']['roofline'], banks['bench_BPSK']['msamples'], 'bpsk_kernel_stats')]
rows = collections.OrderedDict()
traffic = {}


def put(k, v):
    rows.setdefault(k, []).append(v)


for suffix, title, roof, msamples, csvname in cols:
    t = tag + suffix
    ks = newest(f'gpurun_out/{t}/trace/*/*kernel_stats.csv')[0]
    shutil.copy(ks, f'profiles/{rnd}_bench_{csvname}.csv')
    stats = list(csv.DictReader(open(ks)))
    main = max(stats, key=lambda r: float(r['Percentage']))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in newest(f'gpurun_out/{t}/*/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    m = {c: sum(x) / len(x) for c, x in agg[main['Name']].items()}
    if 'k_segf<' in main['Name']:        # round 6: the shift on the filters' side; the masked tail launch stays on k_seg
        tail_name = 'void k_seg<' + main['Name'].split('k_segf<')[1].split(',')[0] + ', 0, -1>(SegArgs)'
    else:
        tail_name = main['Name'].rsplit(',', 1)[0] + ', -1>(SegArgs)'
    tl = {c: sum(x) / len(x) for c, x in agg.get(tail_name, {}).items()}
    bm = 2 * m['FETCH_SIZE'] * 1024 + m['WRITE_SIZE'] * 1024
    bt = 2 * tl.get('FETCH_SIZE', 0) * 1024 + tl.get('WRITE_SIZE', 0) * 1024
    cyc = m['GRBM_GUI_ACTIVE'] / 8
    put('', title)
    put('dominant kernel', '`' + main['Name'].replace('void ', '').replace('(SegArgs)', '').replace('(SegFArgs)', '').replace(', 0, ', ',REDUCE,') + '`')
    put('average launch, kernel trace', f"{float(main['AverageNs']) / 1e3:.1f} us ({main['Percentage']} % of GPU time, {main['Calls']} launches)")
    put('bench line, HIP events', f"{roof['avg_launch_ms'] * 1e3:.1f} us, {msamples} Msamples/s, **{roof['frac']:.3f}** of the fp32 vector peak "
                                  f"({roof.get('frac_r05_formula', roof['frac']):.3f} by the flop count of rounds 1-5)")
    put('nominal flops per launch', f"{roof['flops_per_launch'] / 1e9:.2f} G")
    put('SQ_INSTS_VALU', f"{m['SQ_INSTS_VALU']:.4g}")
    put('GRBM_GUI_ACTIVE / 8 XCDs', f'{cyc:.4g} cycles')
    put('VALU pipe busy = 4 x INSTS_VALU / (1024 SIMDs x cycles)', f"**{4 * m['SQ_INSTS_VALU'] / (1024 * cyc) * 100:.1f} %**")
    put('SQ_WAVE_CYCLES / ACTIVE_INST_ANY / WAIT_ANY / WAIT_INST_ANY',
        f"{m['SQ_WAVE_CYCLES']:.3g} / {m['SQ_ACTIVE_INST_ANY']:.3g} / {m['SQ_WAIT_ANY']:.3g} / {m['SQ_WAIT_INST_ANY']:.3g}")
    put('TCC hit / miss', f"{m['TCC_HIT_sum']:.3g} / {m['TCC_MISS_sum']:.3g}")
    put('FETCH_SIZE (KiB; doubled for bytes) / WRITE_SIZE (KiB)', f"{m['FETCH_SIZE']:.0f} / {m['WRITE_SIZE']:.0f}")
    put('**traffic per launch** = 2 x FETCH + WRITE (+ masked tail kernel)', f'{bm / 1e6:.2f} MB + {bt / 1e6:.2f} = **{(bm + bt) / 1e6:.2f} MB**')
    traffic[suffix] = (int(bm + bt), int(bm), int(bt))
with open(f'profiles/{rnd}_pmc_summary.md', 'w') as f:
    f.write(f"# Round {int(rnd[1:])} -- kernel trace and PMC counters of the bench line's three filter banks at HEAD\n\n"
            f"`tools/prof_bench.sh {tag} | {tag}_cc --protocol CC11xx | {tag}_bpsk --protocol bench_BPSK` on ONE MI355X, right after the un-profiled bench\n"
            f"line `{rnd}_bench.json` on the same box (D = 256, N = 2^20; separate `rocprofv3` passes: `--kernel-trace --stats`, then `--pmc` groups, each under\n"
            f"its own timeout; counters are per dispatch of the dominant kernel, averaged over its dispatches at the settled clock).  Written by\n"
            f"`tools/pmc_summary.py {tag} {rnd}`.\n\n")
    first = True
    for k, v in rows.items():
        f.write('| ' + k + ' | ' + ' | '.join(v) + ' |\n')
        if first:
            f.write('|---|---|---|---|\n')
            first = False
    f.write("\nAll three are compute-bound kernels (`roofline.bound = valu_fp32`); their HBM-side traffic is three orders below the two-pass formulation's\n"
            "34.45 GB per block (SURVEY 8d) because no length-N intermediate exists on this path: the 8.39 MB block once, one float per (bin, filter,\n"
            "slot) of partial sums, the segment spectra -- since round 6 one set per Doppler bin, 4 MiB at C2, 32 MiB for the 384-tap bank -- from L2 (each XCD\n"
            "keeps its eighth of the bins).  `profiles/pmc_traffic.json` carries these traffic figures into `roofline.traffic` of\n"
            f"the bench line (labelled as stored from this profile).  The two-pass fallback has a file of its own (`{rnd}_pmc_twopass.md`); the\n"
            f"2048-point kernel's cost table, A/B runs and taps sweeps are in `r04_long_filter.md`, the costed LDS-DMA variant in `r05_long_filter.md`, round 6's\n"
            f"operation counts, power experiment and the filter-side shift in `r06_fft_ops.md`.\n")
p = 'profiles/pmc_traffic.json'
d = json.load(open(p))
keys = {'': ('segment_D256_M8_N20_L8', 'k_segf<256,13,SUMQ> + masked tail k_seg<256,REDUCE,-1>'), '_cc': ('segment_D256_M8_N20_L11', 'bench.py --protocol CC11xx: k_segf<2048,26,SUMQ> (wave-local) + masked tail'),
        '_bpsk': ('segment_D256_M32_N20_L8', 'bench.py --protocol bench_BPSK: k_segf<256,11,SUMQ> + masked tail; 16 unique filter rows')}
for sfx, (key, what) in keys.items():
    tot, bm, bt = traffic[sfx]
    d[key] = {'bytes': tot, 'main_kernel': bm, 'tail_kernel': bt,
              'source': f'profiles/{rnd}_pmc_summary.md: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of round {int(rnd[1:])} over bench.py ({what}); stored from the profile, not measured by the bench run'}
json.dump(d, open(p, 'w'), indent=1)
print(open(f'profiles/{rnd}_pmc_summary.md').read())
