#!/usr/bin/env python3
"""Plan-quality sweep of the projection planner (imm3_api.cpp: query_create_impl and what it calls).

For every cell -- rows x survivors per row x spread / clustered x SELECT-list shape -- the query is created and run under the planner's
own choice (tuning variant 0, with the sample it takes at creation) and under every forced alternative:
  6 = never the one launch (survivor records -> k_scan -> k_emit where records apply, else the bitmap path),
  3 = no survivor records and no one launch (plain filter -> k_scan -> k_gather),
  8 = the one launch also with gathered SELECT-list columns,  9 = gathered int32 columns streamed through it whatever the selectivity.
Per variant: the kernels of one run (HIP events, median of 10 runs after the host has seen the count once -- steady state) and the
first run of a fresh query (what a one-shot statement pays).  Output: one line per cell, a JSON file, and the share of cells in
which the planner is within 10 % of the best forced plan.
usage: plan_sweep.py [out.json] [rows ...]"""
import json
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth

out_path = sys.argv[1] if len(sys.argv) > 1 else "plan_sweep.json"
sizes = [int(x) for x in sys.argv[2:]] or [4_000_000, 16_000_000, 50_000_000, 100_000_000]
N = max(sizes)
VARIANTS = [0, 6, 3, 8, 9]
GT, LT, MATCH = native.GT, native.LT, native.MATCH
ctx = native.Context(0)
ids = np.arange(N, dtype=np.int32)
age = synth.uniform_below(2, N, 100, np.int8)
st = synth.state_codes(3, N)
codes = [bytes(c) for c in np.unique(st[:100_000], axis=0)]


def segment(n):
    return native.DeviceSegment(ctx, [
        (native.DENSE_INT, 4, ids[:n].view(np.uint8), n * 4, synth.block_offsets(n, 4)),
        (native.DENSE_STRING, 2, st[:n].reshape(-1), n * 2, synth.block_offsets(n, 2)),
        (native.DENSE_TINYINT, 1, age[:n].view(np.uint8), n, synth.block_offsets(n, 1))])


def shapes(n):
    """name -> (sigma label, spread/clustered, used, sels, proj)"""
    out = []
    for pct in (1, 3, 10, 30, 60, 99):
        k = float(pct)                                   # age is uniform on 0..99: age < k keeps k %
        out += [(f"age<{pct} -> age", pct, "spread", [2], [(0, LT, k)], [0]),
                (f"age<{pct} -> id", pct, "spread", [2, 0], [(0, LT, k)], [1]),
                (f"age<{pct} -> id, age", pct, "spread", [2, 0], [(0, LT, k)], [1, 0]),
                (f"age<{pct} and id>=0 -> id, age", pct, "spread", [2, 0], [(0, LT, k), (1, GT, -1.0)], [1, 0]),
                (f"age<{pct} -> state, id, age", pct, "spread", [2, 1, 0], [(0, LT, k)], [1, 2, 0])]
        t = float(int(n * (1.0 - pct / 100.0)))          # id is the sorted key: id > t keeps the last pct % of the rows
        out += [(f"id>{100 - pct}% -> id", pct, "clustered", [0], [(0, GT, t)], [0]),
                (f"id>{100 - pct}% -> id, age", pct, "clustered", [0, 2], [(0, GT, t)], [0, 1]),
                (f"id>{100 - pct}% -> age", pct, "clustered", [0, 2], [(0, GT, t)], [1])]
    for m in (1, 2, 5, 8):                               # state in (m of the 51 codes): m / 51 of the rows
        pct = round(100.0 * m / len(codes), 1)
        lst = codes[:m]
        out += [(f"state in {m} -> id, state, age", pct, "spread", [1, 0, 2], [(0, MATCH, lst)], [1, 0, 2]),
                (f"state in {m} -> age", pct, "spread", [1, 2], [(0, MATCH, lst)], [1]),
                (f"state in {m} -> state", pct, "spread", [1], [(0, MATCH, lst)], [0])]
    return out


def kernels_us(q, reps):
    """median over `reps` runs of the kernels' summed durations (us), and the per-slot medians (filter / scan / project / count)"""
    ctx.sync()
    ctx.timing_enable(512)
    ctx.timing_mask(0xFFFFFFFF)
    ctx.timing_reset()
    for _ in range(reps):
        q.run()
    ctx.sync()
    ks = [ctx.timing_collect(i) for i in range(4)]
    ctx.timing_enable(0)
    per_run = np.zeros(reps)
    slots = []
    for k in ks:
        if k.size:
            r = k.reshape(reps, -1).sum(axis=1)
            per_run += r
            slots.append(round(float(np.median(r)) * 1e3, 1))
        else:
            slots.append(0.0)
    return float(np.median(per_run)) * 1e3, slots


def plan_name(p):
    if p["ran_single_pass"]:
        return f"one launch P={p['P']}"
    return "records" if p["records"] else "bitmap"


cells = []
for n in sizes:
    seg = segment(n)
    for name, pct, kind, used, sels, proj in shapes(n):
        cell = {"rows": n, "shape": name, "sigma_pct": pct, "kind": kind, "variants": {}}
        for v in VARIANTS:
            ctx.set_tuning(v, 0)
            q = native.DeviceQuery(ctx, seg, used, sels, proj, 0, 1024)     # (a first query of this shape and variant: loads the kernels it uses)
            q.run()
            q.close()
            q = native.DeviceQuery(ctx, seg, used, sels, proj, 0, 1024)
            p0 = plan_name({**q.plan(), "ran_single_pass": q.plan()["single_pass"]})
            first, _ = kernels_us(q, 1)                 # the first run of a fresh query: the plan the sample led to
            cnt = q.count()                             # the host has seen the count: later runs may adapt (P, dropped plans)
            for _ in range(2):
                q.run()
            steady, slots = kernels_us(q, 10)
            cell["variants"][str(v)] = {"first_us": round(first, 1), "steady_us": round(steady, 1), "plan": plan_name(q.plan()), "first_plan": p0, "slots_us": slots}
            cell["selected"] = cnt
            q.close()
        ctx.set_tuning(0, 0)
        for key in ("first_us", "steady_us"):
            best = min(cell["variants"][str(v)][key] for v in VARIANTS)
            cell[key.replace("_us", "_ratio")] = round(cell["variants"]["0"][key] / best, 3)
        cells.append(cell)
        vs = cell["variants"]
        print(f"{n:>11,d} {name:34s} sel {cell['selected'] / n:6.3f}  planner: {vs['0']['plan']:18s} first {vs['0']['first_us']:6.1f} steady {vs['0']['steady_us']:6.1f} | "
              + "  ".join(f"v{v} {vs[str(v)]['first_us']:6.1f}/{vs[str(v)]['steady_us']:6.1f} ({vs[str(v)]['plan'][:10]})" for v in VARIANTS[1:])
              + f" | ratio first {cell['first_ratio']:.2f} steady {cell['steady_ratio']:.2f}", flush=True)
    seg.close()
summary = {}
for key in ("first_ratio", "steady_ratio"):
    r = np.array([c[key] for c in cells])
    summary[key] = {"cells": int(r.size), "within_10pct": float((r <= 1.10).mean()), "within_5pct": float((r <= 1.05).mean()), "worst": float(r.max()),
                    "worst_cells": [f"{c['rows']:,d} {c['shape']} ({c[key]:.2f})" for c in sorted(cells, key=lambda c: -c[key])[:8]]}
print(json.dumps(summary, indent=1))
json.dump({"variants": {"0": "planner", "6": "never one launch", "3": "no records, no one launch", "8": "one launch with gathers", "9": "gathered int32 streamed"},
           "summary": summary, "cells": cells}, open(out_path, "w"), indent=0)
