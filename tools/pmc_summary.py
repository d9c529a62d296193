#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per kernel: python tools/pmc_summary.py <counter_collection.csv>"""
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:90], r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
print("kernel,counter,mean_per_dispatch,dispatches")
for (k, c), (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"\"{k}\",{c},{s/n:.3f},{n}")
