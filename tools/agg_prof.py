"""Development tool: a few runs of the group-by query (and the PFOR filter) for rocprofv3 --pmc passes."""
import sys, numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth
n = 100_000_000
ctx = native.Context(0)
ids = np.arange(n, dtype=np.int32); age = synth.uniform_below(2, n, 100, np.int8); st = synth.state_codes(3, n)
seg = native.DeviceSegment(ctx, [(1, 4, ids.view(np.uint8), n*4, synth.block_offsets(n,4)), (3, 2, st.reshape(-1), n*2, synth.block_offsets(n,2)), (2, 1, age.view(np.uint8), n, synth.block_offsets(n,1))])
q = native.DeviceQuery(ctx, seg, [1, 2, 0], [], (), 0, 1024, group_cols=[0], aggs=[(0, 2), (2, 1)])
for _ in range(3): q.run()
ctx.sync(); q.close()
dat, offs = native.pfor_encode_column(ids, 1024)
sp = native.DeviceSegment(ctx, [(native.PFOR_INT, 4, dat, dat.size, offs)])
q = native.DeviceQuery(ctx, sp, [0], [(0, native.GT, 1e6), (0, native.LT, 9e7)])
for _ in range(3): q.run_select()
ctx.sync(); q.close()
