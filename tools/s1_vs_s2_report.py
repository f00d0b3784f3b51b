"""Summary of a tools/s1_vs_s2.sh run: python3 tools/s1_vs_s2_report.py <tag> [out.md]   (reads gpurun_out/<tag>/)"""
import collections
import csv
import glob
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r06_s1s2'
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
root = f'gpurun_out/{tag}'
KERNELS = ('k_segf<256, 13', 'k_seg<256, 0, 13>')       # the search kernel of C2: round 6's form, or (MFB_SEG_FSM=0) the time-side one


def p(*a):
    print(*a, file=out)


def tail_table(path):
    lines = open(path).read().splitlines()
    i = max(k for k, l in enumerate(lines) if 'medians over' in l)
    return lines[i:]


p(f'# S1 against S2 against all-zero samples, one device, one handle (`tools/s1_vs_s2.sh {tag}`)\n')
p('Interleaved legs, each behind 60 ms of untimed steps on its own stimulus, 5 repeats per leg, median of 3 rounds; `search` = HIP events around')
p('the search launches.  S1 = the GMSK bench packet at +fs/4, tiled, AWGN 10 dB; S2 = unit-variance white noise; Z = zeros; S1x2 = S1 at')
p("S2's power; N10 = S1's noise alone; S1c = S1 without noise; S2q = S2 quantised to 8 bits.\n")
for f in ('ab_gmsk.txt', 'ab_cc11xx.txt'):
    try:
        p('```')
        for l in tail_table(f'{root}/{f}'):
            p(l)
        p('```\n')
    except (OSError, ValueError):
        p(f'({f} missing)\n')

p('## Cycles per launch and the clock they ran at\n')
p(f'`rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU` (its own pass) and `--kernel-trace --stats` (its own pass) per')
p(f'stimulus, the search kernel of C2 (`{KERNELS[0]}`; `{KERNELS[1]}` with MFB_SEG_FSM=0).  GRBM_GUI_ACTIVE is summed over the 8 XCDs; cycles = /8; clock = cycles / the un-countered average duration.\n')
p('| stimulus | launches (pmc / trace) | GRBM_GUI_ACTIVE / 8 | SQ_INSTS_VALU | SQ_ACTIVE_INST_VALU | avg duration (trace) | effective clock | VALU busy (4·ACTIVE_INST_VALU / (1024 SIMDs · cycles)) |')
p('|---|---|---|---|---|---|---|---|')
rows = {}
for s in ('S1', 'S2', 'Z'):
    agg = collections.defaultdict(list)
    for f in glob.glob(f'{root}/pmc_{s}/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if any(k in r['Kernel_Name'] for k in KERNELS):
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    dur, calls = None, 0
    for f in glob.glob(f'{root}/trace_{s}/*/*kernel_stats.csv'):
        for r in csv.DictReader(open(f)):
            if any(k in r['Name'] for k in KERNELS):
                dur, calls = float(r['AverageNs']), int(r['Calls'])
    if not agg or dur is None:
        p(f'| {s} | (missing) | | | | | | |')
        continue
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    cyc = m['GRBM_GUI_ACTIVE'] / 8
    ghz = cyc / dur
    busy = 4 * m['SQ_ACTIVE_INST_VALU'] / (1024 * cyc)
    rows[s] = (cyc, dur, ghz)
    p(f"| {s} | {len(agg['GRBM_GUI_ACTIVE'])} / {calls} | {cyc:.4g} | {m['SQ_INSTS_VALU']:.4g} | {m['SQ_ACTIVE_INST_VALU']:.4g} | {dur / 1e3:.1f} µs | "
      f'{ghz:.3f} GHz | {busy:.3f} |')
p('')

p('## Package power and sclk beside a 6000-step run (rocm-smi, 10 samples 0.4 s apart, from second 7 of the process)\n')
p('| stimulus | run | power W (median, min … max) | sclk MHz (median, min … max) | junction °C |')
p('|---|---|---|---|---|')
for s in ('S1', 'S2', 'Z'):
    try:
        txt = open(f'{root}/power_{s}.txt').read()
        run = open(f'{root}/power_run_{s}.log').read().strip().splitlines()[-1]
    except OSError:
        continue
    # a sample taken after the run ended reads idle power: keep samples above 600 W
    pw, ck, tj = [], [], []
    for line in txt.splitlines():
        w = re.search(r'Power \(W\): ([0-9.]+)', line)
        c = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', line)
        t = re.search(r'junction\) \(C\): ([0-9.]+)', line)
        if w and c and float(w.group(1)) > 600:
            pw.append(float(w.group(1)))
            ck.append(float(c.group(1)))
            if t:
                tj.append(float(t.group(1)))
    if not pw:
        continue
    med = lambda v: sorted(v)[len(v) // 2]      # noqa: E731
    p(f'| {s} | {run} | {med(pw):.0f} ({min(pw):.0f} … {max(pw):.0f}; {len(pw)} samples under load) | {med(ck):.0f} ({min(ck):.0f} … {max(ck):.0f}) | '
      f'{med(tj) if tj else float("nan"):.0f} |')
p('')
if 'S1' in rows and 'S2' in rows and 'Z' in rows:
    p(f"Cycles per launch: S2 / S1 = {rows['S2'][0] / rows['S1'][0]:.4f}, Z / S1 = {rows['Z'][0] / rows['S1'][0]:.4f}; "
      f"duration: S2 / S1 = {rows['S2'][1] / rows['S1'][1]:.4f}, Z / S1 = {rows['Z'][1] / rows['S1'][1]:.4f}.")
