#!/usr/bin/env python3
"""Development tool: scan+select kernel time per column-kind combination (100 M rows, HIP-event timing)."""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth

n = 100_000_000
ctx = native.Context(0)
ids = np.arange(n, dtype=np.int32)
v30 = synth.uniform_int30(1, n)
age = synth.uniform_below(2, n, 100, np.int8)
age2 = synth.uniform_below(5, n, 100, np.int8)
st = synth.state_codes(3, n)
seg = native.DeviceSegment(ctx, [
    (1, 4, ids.view(np.uint8), n * 4, synth.block_offsets(n, 4)), (1, 4, v30.view(np.uint8), n * 4, synth.block_offsets(n, 4)),
    (2, 1, age.view(np.uint8), n, synth.block_offsets(n, 1)), (2, 1, age2.view(np.uint8), n, synth.block_offsets(n, 1)),
    (3, 2, st.reshape(-1), n * 2, synth.block_offsets(n, 2))])
GT, LT, MATCH = native.GT, native.LT, native.MATCH
cases = {
    "I32": ([1], [(0, GT, 2.0 ** 28), (0, LT, 3 * 2.0 ** 28)], 4),
    "I8": ([2], [(0, GT, 18.0), (0, LT, 30.0)], 1),
    "S2": ([4], [(0, MATCH, [b"CA"])], 2),
    "S2 in(4)": ([4], [(0, MATCH, [b"CA", b"NY", b"TX", b"WA"])], 2),
    "I32+I32": ([0, 1], [(0, GT, 1e6), (1, LT, 3 * 2.0 ** 28)], 8),
    "I32+I8": ([0, 2], [(0, GT, 1e6), (1, GT, 18.0), (1, LT, 30.0)], 5),
    "I8+I8": ([2, 3], [(0, GT, 18.0), (1, LT, 30.0)], 2),
    "I32+S2": ([0, 4], [(0, GT, 1e6), (1, MATCH, [b"CA"])], 6),
    "I8+S2": ([2, 4], [(0, GT, 18.0), (1, MATCH, [b"CA"])], 3),
    "I32+I8+S2": ([0, 2, 4], [(0, GT, 1e6), (1, GT, 18.0), (2, MATCH, [b"CA"])], 7),
    "I32+I32+I8": ([0, 1, 2], [(0, GT, 1e6), (1, LT, 3 * 2.0 ** 28), (2, GT, 18.0)], 9),
    "none": ([0], [], 0),
}
import os
VARIANT = int(os.environ.get("IMM3_VARIANT", "0"))
grids = [int(g) for g in sys.argv[1:]] or [0]
ONLY = [c for c in os.environ.get("IMM3_KINDS", "").split(",") if c]
if ONLY:
    cases = {k: v for k, v in cases.items() if k in ONLY}
print(f"{'kinds':12s} {'grid':>6s} {'us':>8s} {'GB/s (cols + bitmap)':>22s} {'% of 8 TB/s':>12s}")
for name, (used, sels, bpr) in [(k, v) for k, v in cases.items() for _ in grids]:
    pass
for name, (used, sels, bpr) in cases.items():
  for grid in grids:
    ctx.set_tuning(VARIANT, grid)
    q = native.DeviceQuery(ctx, seg, used, sels)
    for _ in range(3):
        q.run_select()
    ctx.sync()
    ctx.timing_enable(64); ctx.timing_mask(1); ctx.timing_reset()
    for _ in range(20):
        q.run_select()
    ctx.sync()
    ms = float(np.median(ctx.timing_collect(0)))
    ctx.timing_enable(0)
    import time
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(200):
        q.run_select()
    ctx.sync(); step_us = (time.perf_counter() - t0) / 200 * 1e6   # whole select: kernel(s) + count reduce, back to back
    gbs = (bpr + 0.125) * n / (ms * 1e-3) / 1e9
    # the same chain as a count-only run (imm3_query_run_count: no bitmap stored); algorithmic bytes = the columns alone
    for _ in range(3):
        q.run_count()
    ctx.sync()
    ctx.timing_enable(64); ctx.timing_mask(1); ctx.timing_reset()
    for _ in range(20):
        q.run_count()
    ctx.sync()
    cms = float(np.median(ctx.timing_collect(0)))
    ctx.timing_enable(0)
    cgbs = bpr * n / (cms * 1e-3) / 1e9 if bpr else 0.0
    print(f"{name:12s} {grid:6d} {ms * 1e3:8.1f} {gbs:22.0f} {gbs / 80:11.1f}%   step {step_us:6.1f} us   count-only {cms * 1e3:6.1f} us {cgbs:6.0f} GB/s {cgbs / 80:5.1f}%")
    q.close()
