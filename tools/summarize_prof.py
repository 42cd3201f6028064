#!/usr/bin/env python3
"""Copies the rocprofv3 summaries of one profiling session from gpurun_out/ (scratch) into profiles/ (tracked)
and derives profiles/traffic.json (HBM bytes per launch of the dominant kernel) from the PMC passes, corrected
as MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE (KB) counts 64 B per 128-B request on gfx950 for
coalesced streaming reads -> x2 (calibrated here on a known byte count: the kernel must read the whole 400 MB
column and the doubled counter equals it to 0.02 %); WRITE_SIZE (KB) is exact.
usage: tools/summarize_prof.py <tag> [kernel-substring] [rows]"""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
kern = sys.argv[2] if len(sys.argv) > 2 else "k_filter_num"
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000_000
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)
g = os.path.join(root, "gpurun_out")

shutil.copy(os.path.join(g, "prof_trace", "r01_kernel_stats.csv"), os.path.join(out, f"{tag}_kernel_stats.csv"))
log = open(os.path.join(g, "prof_trace.log")).read()
for line in log.splitlines():
    if line.startswith('{"metric"'):
        open(os.path.join(out, f"{tag}_bench_under_rocprof.json"), "w").write(line + "\n")

summary = {}
for name in ("fetch", "write"):
    path = os.path.join(g, f"prof_{name}", "r01_counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(os.path.join(out, f"{tag}_pmc_{name}_summary.csv"), "w") as f:
        f.write("kernel,counter,dispatches,mean,min,max\n")
        for (k, c), v in sorted(agg.items()):
            f.write(f"\"{k}\",{c},{len(v)},{sum(v)/len(v):.4f},{min(v):.4f},{max(v):.4f}\n")
            if kern in k:
                summary[c] = sum(v) / len(v)

fetch_kb, write_kb = summary.get("FETCH_SIZE"), summary.get("WRITE_SIZE")
traffic = {
    "workload": "range_filter_i32", "rows": rows, "kernel": kern, "tag": tag,
    "FETCH_SIZE_KB_raw_mean": fetch_kb, "WRITE_SIZE_KB_raw_mean": write_kb,
    "fetch_bytes_corrected_x2": fetch_kb * 1024 * 2, "write_bytes": write_kb * 1024,
    "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024,
    "correction": "gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies 128-B requests at 64 B -> doubled; WRITE_SIZE exact "
                  "(MI355X_MICROARCH.md, HBM section); separate --pmc passes, one counter each",
}
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
