"""Summarise tools/prof_bench.sh output: kernel stats and per-kernel PMC averages.  usage: prof_report.py <tag>"""
import collections
import csv
import glob
import sys

tag = sys.argv[1]
for f in sorted(glob.glob(f'gpurun_out/{tag}/trace/*/*kernel_stats.csv')):
    print('kernel,calls,avg_us,pct')
    for r in csv.DictReader(open(f)):
        if float(r['Percentage']) > 0.05:
            print(f"{r['Name'][:70]},{r['Calls']},{float(r['AverageNs']) / 1e3:.1f},{r['Percentage']}")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in sorted(glob.glob(f'gpurun_out/{tag}/*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:44]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        disp[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
for k, v in agg.items():
    if not (k.startswith('void k_seg') or k.startswith('void k_pass')):
        continue
    print(k)
    for c, val in sorted(v.items()):
        n = len(disp[(k, c)])
        print(f'   {c:28s} per-dispatch {val / n:14.5g}  ({n} dispatches)')
