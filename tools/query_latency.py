#!/usr/bin/env python3
"""Development tool: wall time per query with NO instrumentation (C3 and C4 at 100 M rows, rows reserved) next to the sum
of the kernels' own durations (HIP events in a second pass): the difference is launch gaps between dependent kernels."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth
n = 100_000_000
ctx = native.Context(0)
ids = np.arange(n, dtype=np.int32); age = synth.uniform_below(2, n, 100, np.int8); st = synth.state_codes(3, n)
seg = native.DeviceSegment(ctx, [(1, 4, ids.view(np.uint8), n*4, synth.block_offsets(n,4)), (3, 2, st.reshape(-1), n*2, synth.block_offsets(n,2)), (2, 1, age.view(np.uint8), n, synth.block_offsets(n,1))])
cases = {"C3": ([2, 0], [(0, native.GT, 18.0), (0, native.LT, 30.0), (1, native.GT, 1e6), (1, native.LT, 9e7)], [1, 0]),
         "C4": ([1, 0, 2], [(0, native.MATCH, [b"CA"])], [1, 0, 2]),
         "C2+project(limit 10)": ([0], [(0, native.GT, 5e7)], [0])}
for name, (used, sels, proj) in cases.items():
    q = native.DeviceQuery(ctx, seg, used, sels, proj, 10 if "limit" in name else 0, 1024)
    q.run(); cnt = q.count(); q.reserve_rows(cnt + 1024)
    for _ in range(5): q.run()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(50): q.run()
    ctx.sync()
    wall = (time.perf_counter() - t0) / 50 * 1e6
    ctx.timing_enable(512); ctx.timing_mask(0xFFFFFFFF); ctx.timing_reset()
    for _ in range(20): q.run()
    ctx.sync()
    k = [float(np.mean(ctx.timing_collect(i))) * 1e3 if ctx.timing_collect(i).size else 0.0 for i in (0, 1, 2, 3)]
    ctx.timing_enable(0)
    print(f"{name:22s} wall {wall:7.1f} us/query   kernels: filter {k[0]:6.1f} scan {k[1]:5.1f} gather {k[2]:6.1f} total {k[3]:4.1f}  sum {sum(k):6.1f} us   gaps {wall - sum(k):5.1f} us")
    q.close()
