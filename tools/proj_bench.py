#!/usr/bin/env python3
"""Development tool: C3 / C4 per-kernel times (HIP events) for a list of filter grids.  usage: proj_bench.py [grid ...]"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth

n = 100_000_000
ctx = native.Context(0)
ids = np.arange(n, dtype=np.int32)
age = synth.uniform_below(2, n, 100, np.int8)
st = synth.state_codes(3, n)
seg = native.DeviceSegment(ctx, [
    (native.DENSE_INT, 4, ids.view(np.uint8), n * 4, synth.block_offsets(n, 4)),
    (native.DENSE_STRING, 2, st.reshape(-1), n * 2, synth.block_offsets(n, 2)),
    (native.DENSE_TINYINT, 1, age.view(np.uint8), n, synth.block_offsets(n, 1))])
cases = {
    "C3": ([2, 0], [(0, native.GT, 18.0), (0, native.LT, 30.0), (1, native.GT, 1e6), (1, native.LT, 9e7)], [1, 0]),
    "C4": ([1, 0, 2], [(0, native.MATCH, [b"CA"])], [1, 0, 2]),
    "age->id": ([2, 0], [(0, native.GT, 18.0), (0, native.LT, 30.0)], [1]),          # projected column is NOT a predicate column
    "age5%->id": ([2, 0], [(0, native.GT, 18.0), (0, native.LT, 24.0)], [1]),
    "age3%->id": ([2, 0], [(0, native.GT, 18.0), (0, native.LT, 22.0)], [1]),
    "age11%->id+age": ([2, 0], [(0, native.GT, 18.0), (0, native.LT, 30.0)], [1, 0]),     # the reference README's example query
    "st10%->age": ([1, 2], [(0, native.MATCH, [b"CA", b"NY", b"TX", b"WA", b"VA"])], [1]),
    "st2%->age": ([1, 2], [(0, native.MATCH, [b"CA"])], [1]),
    "age1%->id": ([2, 0], [(0, native.GT, 98.0)], [1]),
    "age1%->id+age": ([2, 0], [(0, native.GT, 98.0)], [1, 0]),
    "age3%->id+age": ([2, 0], [(0, native.GT, 18.0), (0, native.LT, 22.0)], [1, 0]),
    "st2%->id": ([1, 0], [(0, native.MATCH, [b"CA"])], [1]),
    "st2%->id+st": ([1, 0], [(0, native.MATCH, [b"CA"])], [1, 0]),
    "id2%": ([0], [(0, native.GT, 9.8e7)], [0]),
    "age1%+id": ([2, 0], [(0, native.GT, 98.0), (1, native.GT, 1e6), (1, native.LT, 9e7)], [1, 0]),
    "age3%+id": ([2, 0], [(0, native.GT, 18.0), (0, native.LT, 22.0), (1, native.GT, 1e6), (1, native.LT, 9e7)], [1, 0]),
    "age1%": ([2], [(0, native.GT, 98.0)], [0]),
    "age11%": ([2], [(0, native.GT, 18.0), (0, native.LT, 30.0)], [0]),
    "age30%": ([2], [(0, native.LT, 30.0)], [0]),
    "st2%": ([1], [(0, native.MATCH, [b"CA"])], [0]),
    "st10%": ([1], [(0, native.MATCH, [b"CA", b"NY", b"TX", b"WA", b"VA"])], [0]),
    "age11%+st10%": ([2, 1], [(0, native.GT, 18.0), (0, native.LT, 30.0), (1, native.MATCH, [b"CA", b"NY", b"TX", b"WA", b"VA"])], [0, 1]),
    "age3%": ([2], [(0, native.GT, 18.0), (0, native.LT, 22.0)], [0]),
    "id50%": ([0], [(0, native.GT, 5e7)], [0]),                                       # sigma = 0.5
    "age50%": ([2], [(0, native.LT, 50.0)], [0]),                                     # sigma = 0.5, uniformly spread
    "age30%+id": ([2, 0], [(0, native.GT, 18.0), (0, native.LT, 50.0), (1, native.GT, 1e6), (1, native.LT, 9e7)], [1, 0]),
    "age99%": ([2], [(0, native.LT, 99.0)], [0]),
}
names = {0: "filter", 1: "scan", 2: "project", 3: "count"}
import os
VARIANTS = [int(v) for v in os.environ.get("IMM3_VARIANTS", "0").split(",")]
ONLY = [c for c in os.environ.get("IMM3_CASES", "").split(",") if c]
if ONLY:
    cases = {k: v for k, v in cases.items() if k in ONLY}
grids = [int(g) for g in sys.argv[1:]] or [0]
for name, (used, sels, proj) in cases.items():
  for variant in VARIANTS:
    for grid in grids:
        ctx.set_tuning(variant, grid)
        q = native.DeviceQuery(ctx, seg, used, sels, proj, 0, 1024)
        q.run()
        cnt = q.count()
        if not os.environ.get("IMM3_NO_RESERVE"):
            q.reserve_rows(cnt + 1024)
        for _ in range(3):
            q.run()
        ctx.sync()
        ctx.timing_enable(256); ctx.timing_mask(0xFFFFFFFF); ctx.timing_reset()
        for _ in range(20):
            q.run()
        ctx.sync()
        ks = {i: ctx.timing_collect(i) for i in range(4)}
        ctx.timing_enable(0)
        tot = sum(float(np.median(k)) for k in ks.values() if k.size)
        print(f"{name:8s} v{variant:<3d} grid {grid:5d} sel {cnt / n:.4f}  " + "  ".join(f"{names[i]} {float(np.median(k)) * 1e3:6.1f}" for i, k in ks.items() if k.size) + f"  sum {tot * 1e3:6.1f} us")
        q.close()
