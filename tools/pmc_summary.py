#!/usr/bin/env python3
"""Development tool: mean per-dispatch value of every counter in a rocprofv3 --pmc output directory, per kernel.
usage: pmc_summary.py <dir> [kernel substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if sub in k:
            e = acc[(k, r["Counter_Name"])]; e[0] += 1; e[1] += float(r["Counter_Value"])
for (k, c), (n, v) in sorted(acc.items()):
    print(f"{k[:60]:60s} {c:28s} {n:4d} {v / n:14.4g}")
