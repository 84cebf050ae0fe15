#!/usr/bin/env python3
"""per-step table from a rocprofv3 kernel_stats.csv: stats_table.py <csv> <steps executed> [rows]"""
import csv, re, sys
rows = [(int(r['TotalDurationNs']), int(r['Calls']), r['Name']) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(reverse=True)
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
print("total %.1f us/step, %.1f launches/step" % (sum(r[0] for r in rows) / steps / 1e3, sum(r[1] for r in rows) / steps))
for t, c, n in rows[:top]:
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'^void ', '', n); n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n)
    print("%7.1f us/step %5.1f calls/step avg %6.1f us  %s" % (t / steps / 1e3, c / steps, t / c / 1e3, n[:90]))
