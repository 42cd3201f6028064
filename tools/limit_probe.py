#!/usr/bin/env python3
"""Development tool: kernel time of `select id from t where id > T limit 10` on n sequential rows, as a limit scan (chunks behind a
device-side "limit reached" word) and as a whole select (tuning variant 14).  Event-timed per launch (each HIP event pair has a
floor of ~4 us on this stack, even around a kernel that leaves at once); run under `rocprofv3 --kernel-trace --stats` for the
dispatch durations proper:  usage: limit_probe.py [rows] [T ...]"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
thresholds = [float(x) for x in sys.argv[2:]] or [5.0, n / 2]
reps = 20
ctx = native.Context(0)
ids = np.arange(n, dtype=np.int32)
seg = native.DeviceSegment(ctx, [(native.DENSE_INT, 4, ids.view(np.uint8), n * 4, synth.block_offsets(n, 4))])
for thr in thresholds:
    for variant in (0, 15, 14):
        ctx.set_tuning(variant, 0)            # (chunking is decided when the run is enqueued)
        q = native.DeviceQuery(ctx, seg, [0], [(0, native.GT, thr)], [0], 10)
        for _ in range(3):
            q.run()
        ctx.sync()
        ctx.timing_enable(256)
        ctx.timing_mask(0xFFFFFFFF)
        ctx.timing_reset()
        for _ in range(reps):
            q.run()
        ctx.sync()
        ks = {i: ctx.timing_collect(i) for i in range(4)}
        ctx.timing_enable(0)
        idx, _ = q.fetch_rows()
        assert (idx == np.arange(int(thr) + 1, int(thr) + 11)).all()
        per = {i: float(k.sum()) * 1e3 / reps for i, k in ks.items() if k.size}
        print(f"rows {n} id > {thr:.0f} limit 10, { {0: 'limit scan, fused gather  ', 15: 'limit scan, k_scan + gather', 14: 'whole select               '}[variant] }: "
              + "  ".join(f"slot{i} {v:.1f} us ({ks[i].size // reps} launches)" for i, v in per.items()) + f"  sum {sum(per.values()):.1f} us", flush=True)
        q.close()
        ctx.set_tuning(0, 0)
