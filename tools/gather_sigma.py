"""Development tool: k_gather time vs selectivity (predicate on age, project id + age, 100 M rows)."""
import sys, numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth
n = 100_000_000
ctx = native.Context(0)
ids = np.arange(n, dtype=np.int32); age = synth.uniform_below(2, n, 100, np.int8)
seg = native.DeviceSegment(ctx, [(1, 4, ids.view(np.uint8), n*4, synth.block_offsets(n,4)), (2, 1, age.view(np.uint8), n, synth.block_offsets(n,1))])
for lo, hi in ((19, 21), (18, 24), (18, 30), (18, 45), (18, 70), (-1, 100)):
    for proj, name in (([1, 0], "id+age"), ([1], "id")):
        q = native.DeviceQuery(ctx, seg, [1, 0], [(0, native.GT, float(lo)), (0, native.LT, float(hi))], proj, 0, 1024)
        q.run(); cnt = q.count(); q.reserve_rows(cnt + 1024)
        for _ in range(3): q.run()
        ctx.sync(); ctx.timing_enable(256); ctx.timing_mask(0xFFFFFFFF); ctx.timing_reset()
        for _ in range(10): q.run()
        ctx.sync()
        k = [float(np.median(ctx.timing_collect(i))) * 1e3 for i in (0, 1, 2)]
        ctx.timing_enable(0)
        mb = cnt * (4 + (5 if name == "id+age" else 4)) * 2 / 1e6 - cnt * 4 / 1e6
        print(f"sigma {cnt / n:6.3f} proj {name:7s}: filter {k[0]:6.1f} scan {k[1]:5.1f} gather {k[2]:6.1f} us   gather moves ~{mb:7.1f} MB -> {mb / k[2] / 1e3 * 1e0:5.2f} TB/s")
        q.close()
