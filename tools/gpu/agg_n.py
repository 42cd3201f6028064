import sys, numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth
ctx = native.Context(0)
for n in (1_000_000, 100_000_000):
    ids = np.arange(n, dtype=np.int32); age = synth.uniform_below(2, n, 100, np.int8); st = synth.state_codes(3, n)
    seg = native.DeviceSegment(ctx, [(1, 4, ids.view(np.uint8), n*4, synth.block_offsets(n,4)), (3, 2, st.reshape(-1), n*2, synth.block_offsets(n,2)), (2, 1, age.view(np.uint8), n, synth.block_offsets(n,1))])
    for sels in ([], [(1, native.GT, 18.0), (1, native.LT, 30.0)]):
        for mode in (0, 149, 147):
            ctx.set_tuning(mode, 0)
            q = native.DeviceQuery(ctx, seg, [1, 2, 0], sels, (), 0, 1024, group_cols=[0], aggs=[(0, 2), (2, 1)])
            q.run(); ctx.sync()
            ctx.timing_enable(64); ctx.timing_mask(0xFFFFFFFF); ctx.timing_reset()
            for _ in range(5): q.run()
            ctx.sync()
            k4 = ctx.timing_collect(4)
            print(n, len(sels), mode, "agg launches per run", k4.size // 5, "each", [round(float(x) * 1e3, 1) for x in k4[-(k4.size // 5):]], flush=True)
            ctx.timing_enable(0); q.close()
    seg.close()
ctx.set_tuning(0, 0)
