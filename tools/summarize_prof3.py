#!/usr/bin/env python3
"""(round 3: C3 is one launch, traffic.json carries the kernel name and a hash of its sources)
Copies the rocprofv3 summaries of one profiling session of `bench.py --no-cpu-baseline --no-c5` from gpurun_out/ (scratch)
into profiles/ (tracked) and derives, per BASELINE config, kernel durations, HBM traffic and the fraction of the 8 TB/s peak.

    tools/summarize_prof3.py <tag>        reads gpurun_out/<tag>_trace/, <tag>_fetch/, <tag>_write/ and <tag>_trace.log

FETCH_SIZE (KB) counts 64 B per 128-B request on gfx950 for wide coalesced streaming reads -> x2 (MI355X_MICROARCH.md, HBM
section; calibrated in this session on k_read_stream, which reads exactly 400.0 MB).  For kernels whose reads are NOT wide
streaming reads (k_emit's record pieces and gathers, k_scan) the doubling is uncalibrated: both figures are listed.
WRITE_SIZE (KB) is exact."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
g = os.path.join(root, "gpurun_out")


def find(d, pat):
    hits = glob.glob(os.path.join(g, d, "**", pat), recursive=True)
    if not hits:
        raise SystemExit(f"no {pat} under gpurun_out/{d}")
    return hits[0]


stats_path = find(f"{tag}_trace", "*kernel_stats.csv")
shutil.copy(stats_path, os.path.join(out, f"{tag}_kernel_stats.csv"))
line = None
for l in open(os.path.join(g, f"{tag}_trace.log")):
    if l.startswith('{"metric"'):
        line = json.loads(l)
        open(os.path.join(out, f"{tag}_bench_under_rocprof.json"), "w").write(l)
stats = {r["Name"]: r for r in csv.DictReader(open(stats_path))}

pmc = {}
for name in ("fetch", "write"):
    path = find(f"{tag}_{name}", "*counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(os.path.join(out, f"{tag}_pmc_{name}_summary.csv"), "w") as f:
        f.write("kernel,counter,dispatches,mean,min,max\n")
        for (k, c), v in sorted(agg.items()):
            f.write(f"\"{k}\",{c},{len(v)},{sum(v)/len(v):.4f},{min(v):.4f},{max(v):.4f}\n")
            pmc[(k, c)] = sum(v) / len(v)


def kernel(sub):
    parts = sub.split("*")   # "a*b": a name that contains a, then b
    ks = [k for k in stats if all(x in k for x in parts) and k.find(parts[0]) <= k.find(parts[-1])]
    if len(ks) != 1:
        raise SystemExit(f"kernel '{sub}': {ks}")
    k = ks[0]
    f, w = pmc.get((k, "FETCH_SIZE")), pmc.get((k, "WRITE_SIZE"))
    return {"kernel": k, "launches": int(stats[k]["Calls"]), "avg_us": float(stats[k]["AverageNs"]) / 1e3,
            "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
            "hbm_bytes_fetch_x2": (2 * f * 1024 + w * 1024) if f is not None and w is not None else None,
            "hbm_bytes_fetch_raw": (f * 1024 + w * 1024) if f is not None and w is not None else None}


summary = {"tag": tag, "command": "rocprofv3 --kernel-trace --stats -f csv -- python3 bench.py --no-cpu-baseline --no-c5   (+ --pmc FETCH_SIZE and --pmc WRITE_SIZE passes, --steps 20)",
           "peak_GBps": 8000.0}
head = kernel("k_filter_tile<0, 3, 3, 1, false, true, false>")
calib = kernel("k_read_stream")
summary["c2_headline"] = dict(head, algorithmic_bytes=412.5e6, frac=412.5e6 / (head["avg_us"] * 1e-6) / 8e12,
                              traffic_ratio=head["hbm_bytes_fetch_x2"] / 412.5e6 if head["hbm_bytes_fetch_x2"] else None)
summary["fetch_x2_calibration"] = dict(calib, known_bytes=400.0e6, ratio=calib["hbm_bytes_fetch_x2"] / 400.0e6 if calib["hbm_bytes_fetch_x2"] else None)
# C3 (every SELECT-list column is a predicate column) is ONE launch since round 3: k_filter_project (imm3_project.hip)
one = kernel("k_filter_project<0, 1, 3>")
algo3 = line["extra"]["c3_range_age_id_project"]["algorithmic_bytes"] if line else None
summary["c3_range_age_id_project"] = {"single_pass": one, "kernel_us_sum": one["avg_us"], "algorithmic_bytes": algo3,
                                      "frac": algo3 / (one["avg_us"] * 1e-6) / 8e12 if algo3 else None,
                                      "hbm_bytes_sum_fetch_x2": one["hbm_bytes_fetch_x2"],
                                      "traffic_ratio": one["hbm_bytes_fetch_x2"] / algo3 if one["hbm_bytes_fetch_x2"] and algo3 else None,
                                      "bench_line_frac": line["extra"]["c3_range_age_id_project"]["frac"] if line else None,
                                      "bench_line_kernel_ms": line["extra"]["c3_range_age_id_project"]["kernel_ms"] if line else None,
                                      "note": "the same kernel instance runs the 8 segments of extra.c5 when the bench is not started with --no-c5"}
cfgs = {"c4_match_state_project": ("k_filter_tile<2, 3, 3, *, false, true, true>", "k_emit<1, 2>")}
scan = kernel("k_scan")
for name, (fk, ek) in cfgs.items():
    f, e = kernel(fk), kernel(ek)
    algo = line["extra"][name]["algorithmic_bytes"] if line else None
    total_us = f["avg_us"] + scan["avg_us"] + e["avg_us"]
    traffic = sum(k["hbm_bytes_fetch_x2"] for k in (f, scan, e)) if all(k["hbm_bytes_fetch_x2"] for k in (f, scan, e)) else None
    summary[name] = {"filter_stage": f, "offsets_scan": scan, "emit": e, "kernel_us_sum": total_us, "algorithmic_bytes": algo,
                     "frac": algo / (total_us * 1e-6) / 8e12 if algo else None,
                     "hbm_bytes_sum_fetch_x2": traffic, "traffic_ratio": traffic / algo if traffic and algo else None,
                     "bench_line_frac": line["extra"][name]["frac"] if line else None,
                     "bench_line_kernel_ms": line["extra"][name]["kernel_ms"] if line else None}
agg = kernel("k_group_agg_lanes<1, 1, false, 64>")   # both aggregation configs of the extra block run this kernel (all rows / sigma = 0.11): same cost per tile
summary["agg_group_by_state"] = dict(agg, algorithmic_bytes_all_rows=312.5e6, frac_all_rows=312.5e6 / (agg["avg_us"] * 1e-6) / 8e12,
                                     traffic_ratio_all_rows=agg["hbm_bytes_fetch_x2"] / 312.5e6 if agg["hbm_bytes_fetch_x2"] else None,
                                     bench_line={k: line["extra"][k]["kernel_ms"] for k in ("agg_group_by_state_all_rows", "agg_group_by_state_range_age")} if line else None)
json.dump(summary, open(os.path.join(out, f"{tag}_summary.json"), "w"), indent=1)
t = summary["c2_headline"]
sys.path.insert(0, root)
from bench import headline_source_sha16   # noqa: E402  (no GPU needed: bench.py imports torch lazily)
json.dump({"workload": "range_filter_i32", "rows": 100_000_000, "kernel": t["kernel"], "tag": tag, "source_sha16": headline_source_sha16(),
           "FETCH_SIZE_KB_raw_mean": t["FETCH_SIZE_KB"], "WRITE_SIZE_KB_raw_mean": t["WRITE_SIZE_KB"],
           "hbm_bytes_per_launch": t["hbm_bytes_fetch_x2"],
           "correction": "gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies 128-B requests at 64 B -> doubled; WRITE_SIZE exact "
                         "(MI355X_MICROARCH.md, HBM section); separate --pmc passes, one counter each"},
          open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
