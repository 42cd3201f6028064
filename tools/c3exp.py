import sys, numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth
n = 100_000_000
ctx = native.Context(0)
ids = np.arange(n, dtype=np.int32); age = synth.uniform_below(2, n, 100, np.int8)
seg = native.DeviceSegment(ctx, [(1, 4, ids.view(np.uint8), n*4, synth.block_offsets(n,4)), (2, 1, age.view(np.uint8), n, synth.block_offsets(n,1))])
sels = [(0, native.GT, 18.0), (0, native.LT, 30.0), (1, native.GT, 1e6), (1, native.LT, 9e7)]
for variant, grid in ((3, 0), (0, 512), (0, 1024), (0, 1536), (0, 2048), (4, 512), (4, 1024), (4, 1536)):
    ctx.set_tuning(variant, grid)
    q = native.DeviceQuery(ctx, seg, [1, 0], sels, [1, 0], 0, 1024)
    q.run(); cnt = q.count(); q.reserve_rows(cnt + 1024)
    for _ in range(3): q.run()
    ctx.sync(); ctx.timing_enable(256); ctx.timing_mask(0xFFFFFFFF); ctx.timing_reset()
    for _ in range(20): q.run()
    ctx.sync()
    k = [float(np.median(ctx.timing_collect(i))) * 1e3 for i in (0, 1, 2)]
    ctx.timing_enable(0)
    print(f"variant {variant} grid {grid:5d}: filter {k[0]:6.1f} scan {k[1]:5.1f} gather {k[2]:6.1f} sum {sum(k):6.1f} us")
    q.close()
