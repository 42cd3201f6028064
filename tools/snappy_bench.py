#!/usr/bin/env python3
"""Development tool: one-off GPU decode of snappy-coded columns (k_snappy_decode), 1024-row blocks."""
import sys, time
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
ctx = native.Context(0)
rng = np.random.default_rng(1)
cases = {
    "int32 id // 4 (runs)": (native.SNAPPY_INT, 4, (np.arange(n, dtype=np.int64) // 4).astype(np.int32)),
    "int32 uniform (stored)": (native.SNAPPY_INT, 4, synth.uniform_int30(1, n)),
    "state codes (2 B)": (native.SNAPPY_STRING, 2, synth.state_codes(3, n)),
    "age 0..99 (1 B)": (native.SNAPPY_TINYINT, 1, synth.uniform_below(2, n, 100, np.int8)),
}
print(f"{'column':26s} {'raw MB':>8s} {'stored MB':>10s} {'decode us':>10s} {'GB/s out':>9s} {'encode s':>9s}")
for name, (codec, width, v) in cases.items():
    raw = np.ascontiguousarray(v).view(np.uint8).reshape(-1)
    t0 = time.time()
    parts, offs = [], [0]
    for s in range(0, n, 1024):
        parts.append(native.snappy_encode_block(raw[s * width:(s + 1024) * width]))
        offs.append(offs[-1] + len(parts[-1]))
    enc_s = time.time() - t0
    dat = np.frombuffer(b"".join(parts), dtype=np.uint8)
    seg = native.DeviceSegment(ctx, [(codec, width, dat, dat.size, np.array(offs, dtype=np.int32))])
    ctx.timing_enable(8); ctx.timing_mask(1 << 5); ctx.timing_reset()
    q = native.DeviceQuery(ctx, seg, [0], [], [0], 5)
    ctx.sync()
    ms = float(ctx.timing_collect(5)[0])
    ctx.timing_enable(0)
    q.run(); idx, vals = q.fetch_rows()
    assert vals[0].tobytes() == raw[:5 * width].tobytes()
    print(f"{name:26s} {raw.size / 1e6:8.1f} {dat.size / 1e6:10.1f} {ms * 1e3:10.1f} {raw.size / ms / 1e6:9.1f} {enc_s:9.1f}")
    q.close(); seg.close()
