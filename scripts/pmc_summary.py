#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel (sum over dispatches of the last step)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(k, r['Counter_Name'])] += 1
names = sorted({r['Counter_Name'] for r in rows})
print('kernel'.ljust(60), ' '.join(n[:22].rjust(22) for n in names), ' calls')
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    if len(sys.argv) > 2 and not any(s in k for s in sys.argv[2].split(',')):
        continue
    print(k.ljust(60), ' '.join(('%.4g' % v.get(n, 0)).rjust(22) for n in names), cnt[(k, names[0])])
