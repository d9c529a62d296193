#!/usr/bin/env python3
"""Compact view of a rocprofv3 `*_kernel_stats.csv`: python tools/stats_csv.py <file> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
print(f"{'kernel':72s} {'calls':>7s} {'avg_us':>9s} {'total_ms':>10s} {'pct':>6s}")
for r in rows[:n]:
    print(f"{r['Name'][:72]:72s} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:9.2f} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['Percentage']):6.2f}")
