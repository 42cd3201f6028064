#!/usr/bin/env python3
"""(round 5: C5 is one table query, the README-shaped table rides in the extra block, traffic.json carries one stamped entry per config)
Copies the rocprofv3 summaries of one profiling session of `bench.py --no-cpu-baseline --no-limit` from gpurun_out/ (scratch) into
profiles/ (tracked) and derives, per BASELINE config, kernel durations, HBM traffic and the fraction of the 8 TB/s peak.

    tools/summarize_prof5.py <tag>        reads gpurun_out/<tag>_trace/, <tag>_fetch/, <tag>_write/ and <tag>_trace.log

FETCH_SIZE (KB) counts 64 B per 128-B request on gfx950 for wide coalesced streaming reads -> x2 (MI355X_MICROARCH.md, HBM
section; calibrated in this session on k_read_stream, which reads exactly 400.0 MB).  For kernels whose reads are NOT wide
streaming reads (k_emit's record pieces and gathers, k_scan) the doubling is uncalibrated: both figures are listed.
WRITE_SIZE (KB) is exact.  The table instance of k_filter_project runs two workloads in the bench (C5: 8 x 100 M rows; the
README-shaped table: 100 M rows): their dispatches are told apart by the bytes they fetch."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
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

# per-dispatch durations of the kernel trace (the table instance runs two workloads: split by duration)
trace_path = find(f"{tag}_trace", "*kernel_trace.csv")
durs = collections.defaultdict(list)
for r in csv.DictReader(open(trace_path)):
    durs[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)

raw = {}
for name in ("fetch", "write"):
    path = find(f"{tag}_{name}", "*counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    with open(os.path.join(out, f"{tag}_pmc_{name}_summary.csv"), "w") as f:
        f.write("kernel,counter,dispatches,mean,min,max\n")
        for (k, c), v in sorted(agg.items()):
            f.write(f"\"{k}\",{c},{len(v)},{sum(v)/len(v):.4f},{min(v):.4f},{max(v):.4f}\n")
            raw[(k, c)] = v


def name_of(sub):
    parts = sub.split("*")   # "a*b": a name that contains a, then b
    ks = [k for k in stats if all(x in k for x in parts) and k.find(parts[0]) <= k.find(parts[-1])]
    if len(ks) != 1:
        raise SystemExit(f"kernel '{sub}': {ks}")
    return ks[0]


def mean(v):
    return sum(v) / len(v) if v else None


def kernel(sub, pick=None):
    """pick: None = every dispatch; ("big" | "small", threshold_us) = the dispatches longer / shorter than the threshold (their
    counter values are split at the matching rank: the passes run the same dispatches in the same order)."""
    k = name_of(sub)
    d = durs.get(k, [])
    f, w = raw.get((k, "FETCH_SIZE"), []), raw.get((k, "WRITE_SIZE"), [])
    if pick:
        which, thr = pick
        big_share = sum(1 for x in d if x > thr) / max(len(d), 1)
        d = [x for x in d if (x > thr) == (which == "big")]

        def split(v):
            if not v:
                return v
            s = sorted(v)
            n_big = round(big_share * len(s))
            return s[len(s) - n_big:] if which == "big" else s[: len(s) - n_big]
        f, w = split(f), split(w)
    fm, wm = mean(f), mean(w)
    return {"kernel": k, "launches": len(d), "avg_us": mean(d), "FETCH_SIZE_KB": fm, "WRITE_SIZE_KB": wm,
            "hbm_bytes_fetch_x2": (2 * fm * 1024 + wm * 1024) if fm is not None and wm is not None else None,
            "hbm_bytes_fetch_raw": (fm * 1024 + wm * 1024) if fm is not None and wm is not None else None}


x = line["extra"] if line else {}
summary = {"tag": tag, "command": "rocprofv3 --kernel-trace --stats -f csv -- python3 bench.py --no-cpu-baseline --no-limit   (+ --pmc FETCH_SIZE and --pmc WRITE_SIZE passes, --steps 20)",
           "peak_GBps": 8000.0}
head = kernel("k_filter_tile<0, 3, 3, 1, false, true, false>")
calib = kernel("k_read_stream")
summary["c2_headline"] = dict(head, algorithmic_bytes=412.5e6, frac=412.5e6 / (head["avg_us"] * 1e-6) / 8e12,
                              traffic_ratio=head["hbm_bytes_fetch_x2"] / 412.5e6 if head["hbm_bytes_fetch_x2"] else None)
summary["fetch_x2_calibration"] = dict(calib, known_bytes=400.0e6, ratio=calib["hbm_bytes_fetch_x2"] / 400.0e6 if calib["hbm_bytes_fetch_x2"] else None)


def one_launch(name, sub, pick=None, units=1):
    k = kernel(sub, pick)
    algo = x[name]["algorithmic_bytes"] if name in x and "algorithmic_bytes" in x[name] else (x[name]["roofline"]["algorithmic_bytes_per_query"] * units if name in x else None)
    summary[name] = {"single_pass": k, "kernel_us_sum": k["avg_us"], "algorithmic_bytes": algo,
                     "frac": algo / (k["avg_us"] * 1e-6) / 8e12 if algo and k["avg_us"] else None,
                     "hbm_bytes_sum_fetch_x2": k["hbm_bytes_fetch_x2"],
                     "traffic_ratio": k["hbm_bytes_fetch_x2"] / algo if k["hbm_bytes_fetch_x2"] and algo else None,
                     "frac_traffic": k["hbm_bytes_fetch_x2"] / (k["avg_us"] * 1e-6) / 8e12 if k["hbm_bytes_fetch_x2"] and k["avg_us"] else None}
    return k


c3 = one_launch("c3_range_age_id_project", "k_filter_project<0, 1, 3, false>")
# the table instance: C5's passes (8 x 100 M rows, ~0.9 ms) and the README-shaped table (100 M rows, ~0.12 ms)
c5 = one_launch("c5", "k_filter_project<0, 1, 3, true>", ("big", 400.0), units=8)
rt = one_launch("readme_table_c3", "k_filter_project<0, 1, 3, true>", ("small", 400.0))
cfgs = {"c4_match_state_project": ("k_filter_tile<2, 3, 3, *, false, false, true>", "k_emit<1, 2>")}
scan = kernel("k_scan")
for name, (fk, ek) in cfgs.items():
    f, e = kernel(fk), kernel(ek)
    algo = x[name]["algorithmic_bytes"] if name in x else None
    total_us = f["avg_us"] + scan["avg_us"] + e["avg_us"]
    traffic = sum(k["hbm_bytes_fetch_x2"] for k in (f, scan, e)) if all(k["hbm_bytes_fetch_x2"] for k in (f, scan, e)) else None
    summary[name] = {"filter_stage": f, "offsets_scan": scan, "emit": e, "kernel_us_sum": total_us, "algorithmic_bytes": algo,
                     "frac": algo / (total_us * 1e-6) / 8e12 if algo else None,
                     "hbm_bytes_sum_fetch_x2": traffic, "traffic_ratio": traffic / algo if traffic and algo else None,
                     "frac_traffic": traffic / (total_us * 1e-6) / 8e12 if traffic else None,
                     "bench_line_frac": x[name]["frac"] if name in x else None}
agg = kernel("k_group_agg_lanes<1, 1, false, 64, true>")   # both aggregation configs of the extra block run this kernel, the select fused in
summary["agg_group_by_state"] = dict(agg, algorithmic_bytes_all_rows=312.5e6, frac_all_rows=312.5e6 / (agg["avg_us"] * 1e-6) / 8e12,
                                     traffic_ratio_all_rows=agg["hbm_bytes_fetch_x2"] / 312.5e6 if agg["hbm_bytes_fetch_x2"] else None,
                                     bench_line={k: x[k]["kernel_ms"] for k in ("agg_group_by_state_all_rows", "agg_group_by_state_range_age")} if x else None)
json.dump(summary, open(os.path.join(out, f"{tag}_summary.json"), "w"), indent=1)
t = summary["c2_headline"]
sys.path.insert(0, root)
from bench import extra_source_sha16, headline_source_sha16   # noqa: E402  (no GPU needed: bench.py imports torch lazily)
esha = extra_source_sha16()
corr = ("gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 B tallies 128-B requests at 64 B -> doubled; WRITE_SIZE exact "
        "(MI355X_MICROARCH.md, HBM section); separate --pmc passes, one counter each")
json.dump({"workload": "range_filter_i32", "rows": 100_000_000, "kernel": t["kernel"], "tag": tag, "source_sha16": headline_source_sha16(),
           "FETCH_SIZE_KB_raw_mean": t["FETCH_SIZE_KB"], "WRITE_SIZE_KB_raw_mean": t["WRITE_SIZE_KB"],
           "hbm_bytes_per_launch": t["hbm_bytes_fetch_x2"], "correction": corr,
           "extra": {
               "c3_range_age_id_project": {"kernels": [c3["kernel"]], "hbm_bytes_per_query": c3["hbm_bytes_fetch_x2"], "source_sha16": esha, "tag": tag},
               "c4_match_state_project": {"kernels": [summary["c4_match_state_project"][k]["kernel"] for k in ("filter_stage", "offsets_scan", "emit")],
                                          "hbm_bytes_per_query": summary["c4_match_state_project"]["hbm_bytes_sum_fetch_x2"], "source_sha16": esha, "tag": tag,
                                          "note": "the doubling of FETCH_SIZE is calibrated on wide streaming reads; k_emit's gathers fetch whole 128-byte lines (DESIGN finding 34)"},
               "c5_table": {"kernels": [c5["kernel"]], "hbm_bytes_per_pass": c5["hbm_bytes_fetch_x2"], "segments_per_pass": 8, "source_sha16": esha, "tag": tag},
               "readme_table_c3": {"kernels": [rt["kernel"]], "hbm_bytes_per_query": rt["hbm_bytes_fetch_x2"], "source_sha16": esha, "tag": tag},
           }},
          open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
