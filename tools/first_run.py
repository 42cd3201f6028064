#!/usr/bin/env python3
"""Development tool: what ONE-SHOT projecting queries cost (the reference's Engine plans, runs and drops a pipeline per statement):
query creation (wall clock; with and without the sampled selectivity estimate) and the kernels of the FIRST run (HIP events).
usage: first_run.py   (IMM3_CASES=a,b to pick)"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth

n = 100_000_000
ctx = native.Context(0)
ids = np.arange(n, dtype=np.int32)
age = synth.uniform_below(2, n, 100, np.int8)
seg = native.DeviceSegment(ctx, [
    (native.DENSE_INT, 4, ids.view(np.uint8), n * 4, synth.block_offsets(n, 4)),
    (native.DENSE_TINYINT, 1, age.view(np.uint8), n, synth.block_offsets(n, 1))])
GT, LT = native.GT, native.LT
cases = {
    "C3 (10 %)": ([1, 0], [(0, GT, 18.0), (0, LT, 30.0), (1, GT, 1e6), (1, LT, 9e7)], [1, 0]),
    "age in (18,30) -> id, age": ([1, 0], [(0, GT, 18.0), (0, LT, 30.0)], [1, 0]),
    "age < 50 -> age": ([1], [(0, LT, 50.0)], [0]),
    "age in (18,50), id range -> id, age": ([1, 0], [(0, GT, 18.0), (0, LT, 50.0), (1, GT, 1e6), (1, LT, 9e7)], [1, 0]),
    "id > 5e7 -> id": ([0], [(0, GT, 5e7)], [0]),
    "age > 97 -> id, age (2 %)": ([1, 0], [(0, GT, 97.0)], [1, 0]),
}
only = [c for c in os.environ.get("IMM3_CASES", "").split(",") if c]
names = {0: "filter", 1: "scan", 2: "project", 3: "count"}
warm = native.DeviceQuery(ctx, seg, [1], [(0, GT, 120.0)], [0], 0)   # (the pool and the kernels' code objects are warm for everybody)
warm.run(); warm.count(); warm.close()
for name, (used, sels, proj) in cases.items():
    if only and name not in only:
        continue
    for variant in (10, 0):
        ctx.set_tuning(variant, 0)
        ctx.sync()
        t0 = time.perf_counter()
        q = native.DeviceQuery(ctx, seg, used, sels, proj, 0)
        t1 = time.perf_counter()
        ctx.set_tuning(0, 0)
        plan0 = q.plan()
        ctx.timing_enable(64); ctx.timing_mask(0xFFFFFFFF); ctx.timing_reset()
        q.run()
        ctx.sync()
        ks = {i: ctx.timing_collect(i) for i in range(4)}
        ctx.timing_enable(0)
        tot = sum(float(k.sum()) for k in ks.values() if k.size)
        cnt = q.count()
        print(f"{name:38s} {'sampled' if variant == 0 else 'no sample':9s} create {1e3 * (t1 - t0):6.2f} ms  plan one-launch={plan0['single_pass']!s:5s} P={plan0['P']:2d}  first run: "
              + "  ".join(f"{names[i]} {float(k.sum()) * 1e3:6.1f}" for i, k in ks.items() if k.size) + f"  = {tot * 1e3:6.1f} us   sel {cnt / n:.3f}", flush=True)
        q.close()
