"""Development tool: host-side cost of a table query (98 README-style segments, 100 M rows): create / run / count / close."""
import sys, time, numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth
ctx = native.Context(0)
segs = []
rows = 1024 * 1000 + 1
for s in range(98):
    n = rows
    age = synth.uniform_below(100 + s, n, 100, np.int8)
    offs = np.concatenate([np.arange(0, 1001) * 1024, [n]]).astype(np.int32)
    segs.append(native.DeviceSegment(ctx, [(native.DENSE_TINYINT, 1, age.view(np.uint8), n, offs)]))
t = native.DeviceTable(ctx, segs)
sels = [(0, native.GT, 18.0), (0, native.LT, 30.0)]
for rep in range(3):
    t0 = time.perf_counter(); q = native.DeviceQuery(ctx, t, [0], sels); t1 = time.perf_counter()
    q.run_select(); t2 = time.perf_counter(); c = q.count(); t3 = time.perf_counter(); q.close(); t4 = time.perf_counter()
    print(f"create {1e6*(t1-t0):7.1f} us  run(enqueue) {1e6*(t2-t1):6.1f} us  count(sync) {1e6*(t3-t2):6.1f} us  close {1e6*(t4-t3):6.1f} us  count={c}")
