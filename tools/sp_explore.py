#!/usr/bin/env python3
"""Development tool: the single-pass projection kernel (k_filter_project) under its ablation switches and tunings.
Run with IMM3_LIB_PATH=immutable3_amd/lib/libimm3_ablate.so (make -C immutable3_amd/csrc ablate).
usage: sp_explore.py case variant[:grid] ..."""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from immutable3_amd import native, synth

n = 100_000_000
ctx = native.Context(0)
ids = np.arange(n, dtype=np.int32)
age = synth.uniform_below(2, n, 100, np.int8)
st = synth.state_codes(3, n)
seg = native.DeviceSegment(ctx, [
    (native.DENSE_INT, 4, ids.view(np.uint8), n * 4, synth.block_offsets(n, 4)),
    (native.DENSE_STRING, 2, st.reshape(-1), n * 2, synth.block_offsets(n, 2)),
    (native.DENSE_TINYINT, 1, age.view(np.uint8), n, synth.block_offsets(n, 1))])
cases = {
    "C3": ([2, 0], [(0, native.GT, 18.0), (0, native.LT, 30.0), (1, native.GT, 1e6), (1, native.LT, 9e7)], [1, 0]),
    "C4": ([1, 0, 2], [(0, native.MATCH, [b"CA"])], [1, 0, 2]),
    "C4-id-age": ([1, 0, 2], [(0, native.MATCH, [b"CA"])], [1, 2]),       # (the same without the matched column in the SELECT list)
    "age->id": ([2, 0], [(0, native.GT, 18.0), (0, native.LT, 30.0)], [1]),
    "id2%": ([0], [(0, native.GT, 9.8e7)], [0]),
    "id50%": ([0], [(0, native.GT, 5e7)], [0]),
}
case = sys.argv[1]
used, sels, proj = cases[case]
for spec in sys.argv[2:]:
    parts = spec.split(":")
    variant = int(parts[0])
    grid = int(parts[1]) if len(parts) > 1 else 0
    create_variant = int(parts[2]) if len(parts) > 2 else (variant if variant > 200 or variant == 6 else 0)
    ctx.set_tuning(create_variant, grid)
    q = native.DeviceQuery(ctx, seg, used, sels, proj, 0, 1024)
    ctx.set_tuning(variant, grid)
    for _ in range(3):
        q.run()
    ctx.sync()
    plan = q.plan()
    ctx.timing_enable(256); ctx.timing_mask(0xFFFFFFFF); ctx.timing_reset()
    for _ in range(20):
        q.run()
    ctx.sync()
    ks = {i: ctx.timing_collect(i) for i in range(4)}
    ctx.timing_enable(0)
    tot = sum(float(np.median(k)) for k in ks.values() if k.size)
    print(f"{case:8s} variant {variant:4d} grid {grid:5d} create {create_variant:4d}: " + "  ".join(f"k{i} {float(np.median(k)) * 1e3:6.1f}" for i, k in ks.items() if k.size) + f"  sum {tot * 1e3:6.1f} us  P {plan['P']} grid {plan['grid']} sp {plan['ran_single_pass']}", flush=True)
    q.close()
