import sys, time, numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth
n = 100_000_000
ctx = native.Context(0)
ids = np.arange(n, dtype=np.int32); age = synth.uniform_below(2, n, 100, np.int8); st = synth.state_codes(3, n)
seg = native.DeviceSegment(ctx, [(1, 4, ids.view(np.uint8), n*4, synth.block_offsets(n,4)), (3, 2, st.reshape(-1), n*2, synth.block_offsets(n,2)), (2, 1, age.view(np.uint8), n, synth.block_offsets(n,1))])
modes = [int(m) for m in sys.argv[1:]] or [0, 101, 102]
for sels in ([], [(1, native.GT, 18.0), (1, native.LT, 30.0)]):
    for mode in modes:
        ctx.set_tuning(mode, 0)
        q = native.DeviceQuery(ctx, seg, [1, 2, 0], sels, (), 0, 1024, group_cols=[0], aggs=[(0, 2), (2, 1)])
        q.run(); ctx.sync()
        ctx.timing_enable(64); ctx.timing_mask(0xFFFFFFFF); ctx.timing_reset()
        for _ in range(5): q.run()
        ctx.sync()
        print(len(sels), mode, round(float(np.mean(ctx.timing_collect(4))) * 1e3, 1), "us")
        ctx.timing_enable(0); q.close()
