import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from immutable3_amd import native, synth
from conftest import RawColumn, blocks_of, DENSE_INT, GT, LT
ctx = native.Context(0)
n = 3_000_000
a = synth.uniform_int30(1, n); b = synth.uniform_int30(2, n); c = synth.uniform_below(3, n, 100, np.int8)
br = blocks_of(n, 1024)
seg = native.DeviceSegment(ctx, [RawColumn(DENSE_INT, 4, a, br).native(), RawColumn(DENSE_INT, 4, b, br).native(), RawColumn(2, 1, c, br).native()])
for name, sels, keep in (
    ("I32+I32 ~100%", [(0, GT, -1.0), (1, GT, -1.0)], np.ones(n, bool)),
    ("I32+I32 75%", [(0, GT, float(2**27)), (1, GT, float(2**27))], (a > 2**27) & (b > 2**27)),
    ("I32+I32+I8 ~100%", [(0, GT, -1.0), (1, GT, -1.0), (2, GT, -1.0)], np.ones(n, bool)),
    ("I32+I8 ~100%", [(0, GT, -1.0), (2, GT, -1.0)], np.ones(n, bool)),
    ("I32 100%", [(0, GT, -1.0)], np.ones(n, bool))):
    used = sorted({s[0] for s in sels})
    remap = {u: i for i, u in enumerate(used)}
    q = native.DeviceQuery(ctx, seg, used, [(remap[s[0]], s[1], s[2]) for s in sels], list(range(len(used))), 0)
    q.run()
    idx, vals = q.fetch_rows()
    rows = np.flatnonzero(keep)
    ok = idx.size == rows.size and (idx == rows).all()
    cols = [a, b, c]
    for j, u in enumerate(used):
        v = vals[j].view("<i4" if u < 2 else np.int8).reshape(-1)
        ok = ok and (v == cols[u][rows]).all()
    print(name, "rows", rows.size, "OK" if ok else "MISMATCH")
    q.close()
