#!/usr/bin/env python3
"""Development tool: per-work-group timeline of one k_filter_project launch (tools' build: IMM3_LIB_PATH=.../libimm3_ablate.so).
usage: sp_timeline.py variant [create_variant]"""
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
variant = int(sys.argv[1])
create = int(sys.argv[2]) if len(sys.argv) > 2 else 0
case = sys.argv[3] if len(sys.argv) > 3 else "C3"
ctx.set_tuning(create, 0)
if case == "C4":
    st = synth.state_codes(3, n)
    seg = native.DeviceSegment(ctx, [(native.DENSE_INT, 4, ids.view(np.uint8), n * 4, synth.block_offsets(n, 4)),
                                     (native.DENSE_STRING, 2, st.reshape(-1), n * 2, synth.block_offsets(n, 2)),
                                     (native.DENSE_TINYINT, 1, age.view(np.uint8), n, synth.block_offsets(n, 1))])
    q = native.DeviceQuery(ctx, seg, [1, 0, 2], [(0, native.MATCH, [b"CA"])], [1, 0, 2], 0, 1024)
elif case == "id50%":
    q = native.DeviceQuery(ctx, seg, [0], [(0, native.GT, 5e7)], [0], 0, 1024)
else:
    q = native.DeviceQuery(ctx, seg, [1, 0], [(0, native.GT, 18.0), (0, native.LT, 30.0), (1, native.GT, 1e6), (1, native.LT, 9e7)], [1, 0], 0, 1024)
ctx.set_tuning(variant, 0)
for _ in range(3):
    q.run()
ctx.sync()
ctx.devclock_enable(2)
q.run()
ctx.sync()
plan = q.plan()
g = plan["grid"]
raw = ctx.devclock_raw(0, 27 * 256 + 64).astype(np.int64)
t0 = raw[0:2 * g:2][raw[0:2 * g:2] > 0].min()
start = (raw[0:2 * g:2] - t0) / 100.0
end = (raw[1:2 * g:2] - t0) / 100.0
print(f"plan {plan}")
print(f"wg start us: min {start.min():.1f} p50 {np.median(start):.1f} max {start.max():.1f};  end us: min {end.min():.1f} p50 {np.median(end):.1f} max {end.max():.1f}")
ext = raw[2 * g: 2 * g + 24 * g].reshape(g, 24)
sd = raw[26 * g: 27 * g]
if (sd > 0).any():
    sdu = (sd[sd > 0] - t0) / 100.0
    print(f"streamers through by us: min {sdu.min():.1f} p50 {np.median(sdu):.1f} max {sdu.max():.1f};  work-group end - streamers through: p50 {np.median(end[sd > 0] - sdu):.1f} max {(end[sd > 0] - sdu).max():.1f}")
if (sd > 0).all():
    sdu_all = (sd - t0) / 100.0
    print("streamers through, mean us by blockIdx % 8 (XCD under round-robin placement):", " ".join(f"{x}:{sdu_all[x::8].mean():.1f}" for x in range(8)))
    print("streamers through, mean us by blockIdx // 32:", " ".join(f"{x}:{sdu_all[32 * x: 32 * x + 32].mean():.1f}" for x in range(g // 32)))
    srt = np.sort(sdu_all)
    print("streamers through, percentiles us:", " ".join(f"p{p}:{srt[int(p / 100 * (g - 1))]:.1f}" for p in (0, 10, 25, 50, 75, 90, 100)))
for i in range(6):
    r = ext[:, i]; d = ext[:, 6 + i]; pk = ext[:, 12 + i]; an = ext[:, 18 + i]
    if (an > 0).any():
        aa = (an[an > 0] - t0) / 100.0
        print(f"range {i}: announce begun by us min {aa.min():6.1f} p50 {np.median(aa):6.1f} max {aa.max():6.1f}")
    if (r > 0).any():
        rr = (r[r > 0] - t0) / 100.0; dd = (d[d > 0] - t0) / 100.0; pp = (pk[pk > 0] - t0) / 100.0
        if pp.size:
            print(f"range {i}: first row known by us min {pp.min():6.1f} p50 {np.median(pp):6.1f} max {pp.max():6.1f}")
        print(f"range {i}: streamed by us min {rr.min():6.1f} p50 {np.median(rr):6.1f} max {rr.max():6.1f}   drained by us min {dd.min() if dd.size else 0:6.1f} p50 {np.median(dd) if dd.size else 0:6.1f} max {dd.max() if dd.size else 0:6.1f}")
order = np.argsort(start)
print("start us by blockIdx (every 32nd):", " ".join(f"{b}:{start[b]:.0f}" for b in range(0, g, 32)))
print("sorted starts (every 32nd):", " ".join(f"{start[order[k]]:.0f}" for k in range(0, g, 32)))
print("end   us by blockIdx (every 32nd):", " ".join(f"{b}:{end[b]:.0f}" for b in range(0, g, 32)))
