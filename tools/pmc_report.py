"""Summarise rocprofv3 outputs under gpurun_out/<tag>/ per kernel."""
import csv, collections, glob, sys
tag = sys.argv[1]
kpat = sys.argv[2] if len(sys.argv) > 2 else 'k_pass'
for f in sorted(glob.glob(f'gpurun_out/{tag}/trace/*/*kernel_stats.csv')):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage']) > 0.5:
            print(f"{r['Name'][:48]:48s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:10.1f} us  {r['Percentage']}%")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in sorted(glob.glob(f'gpurun_out/{tag}/pmc*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        disp[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
for k, v in agg.items():
    if kpat not in k:
        continue
    print(k)
    for c, val in sorted(v.items()):
        n = len(disp[(k, c)])
        print(f'   {c:24s} total {val:12.4g}   per-dispatch {val/n:12.4g}  ({n} dispatches)')
