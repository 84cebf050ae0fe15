#!/usr/bin/env python3
"""Per-launch durations of the MFMA kernels for the last full step in a rocprofv3 kernel trace."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step starts at stem_fwd
starts = [i for i, r in enumerate(rows) if 'stem_fwd' in r['Kernel_Name']]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]['Start_Timestamp'])
tot = {}
for r in rows[a:b]:
    nm = r['Kernel_Name']
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    short = nm.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if short.startswith('_ZN'):
        import re
        m = re.match(r'_ZN12_GLOBAL__N_1\d+([a-z_0-9A-Z]+?)I(.*?)EEv', short)
        short = (m.group(1) + '<' + m.group(2) + '>') if m else short[:50]
    tot[short] = tot.get(short, 0) + d
    if len(sys.argv) > 2 and any(k in nm for k in sys.argv[2].split(',')):
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f}us {d:8.1f}us grid={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):>6} {short[:70]}")
print('--- per-step totals (us), step span %.1f us' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"{v:9.1f}  {k[:100]}")
print('sum %.1f' % sum(tot.values()))
