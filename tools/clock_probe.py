#!/usr/bin/env python3
"""Development tool (tools' build: IMM3_LIB_PATH=.../libimm3_ablate.so): the shader clock the chip holds under the plain scan+select
kernel and under the one-launch projection kernel -- shader cycles (s_memtime) over device time (the 100 MHz counter), per work-group."""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth

n = 100_000_000
ctx = native.Context(0)
ids = np.arange(n, dtype=np.int32)
age = synth.uniform_below(2, n, 100, np.int8)
seg = native.DeviceSegment(ctx, [(native.DENSE_INT, 4, ids.view(np.uint8), n * 4, synth.block_offsets(n, 4)),
                                 (native.DENSE_TINYINT, 1, age.view(np.uint8), n, synth.block_offsets(n, 1))])
sels = [(0, native.GT, 18.0), (0, native.LT, 30.0), (1, native.GT, 1e6), (1, native.LT, 9e7)]
for name, proj, slot in (("k_filter_tile<I32,I8> (select only)", [], 2), ("k_filter_project<I32,I8> (C3)", [1, 0], 27)):
    q = native.DeviceQuery(ctx, seg, [1, 0], sels, proj, 0, 1024)
    for _ in range(5):
        q.run()
    ctx.sync()
    ctx.devclock_enable(2)
    q.run()
    ctx.sync()
    g = q.plan()["grid"] if proj else 512
    raw = ctx.devclock_raw(0, 28 * 512).astype(np.int64)
    t = (raw[1:2 * g:2] - raw[0:2 * g:2]) / 100.0            # us per work-group
    cyc = raw[slot * g: slot * g + g]
    ok = (t > 0) & (cyc > 0)
    print(f"{name}: work-groups {g}, lifetime p50 {np.median(t[ok]):.1f} us, shader clock p50 {np.median(cyc[ok] / t[ok]) / 1e3:.3f} GHz (min {np.min(cyc[ok] / t[ok]) / 1e3:.3f}, max {np.max(cyc[ok] / t[ok]) / 1e3:.3f})")
    ctx.devclock_enable(0)
    q.close()
