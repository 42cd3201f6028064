#!/usr/bin/env python3
"""Development tool (round 5): what does the one launch cost over a TABLE (k_filter_project's table instances) against the same
rows as one segment / as one launch per segment?  C3's query over 100 M rows cut four ways.  usage: table_probe.py [rows]"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ctx = native.Context(0)
ids = np.arange(n, dtype=np.int32)
age = synth.uniform_below(2, n, 100, np.int8)
lo, hi = 1e6 * n / 1e8, 9e7 * n / 1e8
sels = [(0, native.GT, 18.0), (0, native.LT, 30.0), (1, native.GT, lo), (1, native.LT, hi)]
want = int(((age > 18) & (age < 30) & (ids > lo) & (ids < hi)).sum())


def segments(bounds):
    out = []
    for a, b in bounds:
        m = b - a
        out.append(native.DeviceSegment(ctx, [(native.DENSE_TINYINT, 1, age[a:b].view(np.uint8), m, synth.block_offsets(m, 1)),
                                              (native.DENSE_INT, 4, ids[a:b].view(np.uint8), m * 4, synth.block_offsets(m, 4))]))
    return out


def kernel_us(queries, reps=20):
    for q in queries:
        q.run()
    total = sum(q.count() for q in queries)
    assert total == want, (total, want)
    for q in queries:
        q.reserve_rows(q.count() + 1024)
    for _ in range(3):
        for q in queries:
            q.run()
    ctx.sync()
    ctx.timing_enable(8 * len(queries) * reps + 64); ctx.timing_mask(0xFFFFFFFF); ctx.timing_reset()
    for _ in range(reps):
        for q in queries:
            q.run()
    ctx.sync()
    per = np.zeros(reps)
    launches = 0
    for i in range(4):
        k = ctx.timing_collect(i)
        if k.size:
            per += k.reshape(reps, -1).sum(axis=1)
            launches += k.size // reps
    ctx.timing_enable(0)
    plans = [q.plan() for q in queries]
    return float(np.median(per)) * 1e3, launches, plans[0]


def cut(rows_per_segment):
    return [(a, min(a + rows_per_segment, n)) for a in range(0, n, rows_per_segment)]


cases = [("1 segment x %d rows" % n, [(0, n)]),
         ("1 segment, whole tiles only", [(0, n // 1024 * 1024)]),
         ("8 segments", cut((n + 7) // 8)),
         ("8 segments of whole tiles", cut(((n + 7) // 8 + 1023) // 1024 * 1024)),
         ("98 loader-made segments (1000 * 1024 + 1 rows)", cut(1000 * 1024 + 1)),
         ("98 segments of 1000 * 1024 rows", cut(1000 * 1024))]
for name, bounds in cases:
    if bounds[-1][1] != n:   # (a case that drops the tail rows: its own expected count)
        m = bounds[-1][1]
        want_here = int(((age[:m] > 18) & (age[:m] < 30) & (ids[:m] > lo) & (ids[:m] < hi)).sum())
    else:
        want_here = want
    segs = segments(bounds)
    keep_want, want = want, want_here
    per_seg = [native.DeviceQuery(ctx, s, [0, 1], sels, [1, 0], 0, 1024) for s in segs] if len(segs) <= 8 else None
    line = f"{name:48s}"
    if per_seg:
        us, launches, plan = kernel_us(per_seg)
        line += f"  per-segment queries {us:7.1f} us ({launches} launches, one launch: {plan['single_pass']}, P {plan['P']})"
        for q in per_seg:
            q.close()
    table = native.DeviceTable(ctx, segs)
    q = native.DeviceQuery(ctx, table, [0, 1], sels, [1, 0], 0, 1024)
    us, launches, plan = kernel_us([q])
    line += f"  table query {us:7.1f} us ({launches} launch(es), one launch: {plan['single_pass']}, P {plan['P']}, spans {plan['spans']})"
    q.close()
    for P in (4, 6, 8):
        ctx.set_tuning(200 + P, 0)
        q = native.DeviceQuery(ctx, table, [0, 1], sels, [1, 0], 0, 1024)
        ctx.set_tuning(0, 0)
        us, _, _ = kernel_us([q])
        line += f"  P={P}: {us:6.1f}"
        q.close()
    print(line, flush=True)
    want = keep_want
    table.close()
    for s in segs:
        s.close()
