import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from immutable3_amd import native, synth
n = 100_000_000
ctx = native.Context(0)
ids = np.arange(n, dtype=np.int32); age = synth.uniform_below(2, n, 100, np.int8); st = synth.state_codes(3, n)
seg = native.DeviceSegment(ctx, [(1, 4, ids.view(np.uint8), n*4, synth.block_offsets(n,4)), (3, 2, st.reshape(-1), n*2, synth.block_offsets(n,2)), (2, 1, age.view(np.uint8), n, synth.block_offsets(n,1))])
for mode in (0, 101, 102):
    ctx.set_tuning(mode, 0)
    q = native.DeviceQuery(ctx, seg, [1, 2, 0], [], (), 0, 1024, group_cols=[0], aggs=[(0, 2), (2, 1)])
    q.run(); ctx.sync()
    ctx.timing_enable(64); ctx.timing_mask(0xFFFFFFFF); ctx.timing_reset()
    for _ in range(5): q.run()
    ctx.sync()
    print(mode, float(np.mean(ctx.timing_collect(4))))
    ctx.timing_enable(0); q.close()
