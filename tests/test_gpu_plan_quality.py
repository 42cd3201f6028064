"""Plan quality on the device: for a handful of cells of tools/plan_sweep.py's grid -- 16 M rows, few /
some / most survivors, spread and clustered, four SELECT-list shapes -- the plan the library picks (cost model, csrc/imm3_plan.h, on
the sample taken at creation and then on the first count) runs within 25 % of the best plan the tuning hook can force (never the one
launch, no records, the one launch with gathers, gathered int32 streamed; each measured twice, the faster run counts).  The committed sweep (profiles/r04_plan_sweep.*) is the
full grid -- 240 cells, 97.5 % within 10 % -- and the tighter bound; this is its regression test, loose enough for HIP-event noise
on 20-60 us queries."""
import numpy as np
import pytest

from immutable3_amd import native, synth

pytestmark = pytest.mark.gpu
GT, LT, MATCH = native.GT, native.LT, native.MATCH
N = 16_000_000


def kernels_us(ctx, q, reps=10):
    ctx.sync()
    ctx.timing_enable(512)
    ctx.timing_mask(0xFFFFFFFF)
    ctx.timing_reset()
    for _ in range(reps):
        q.run()
    ctx.sync()
    per_run = np.zeros(reps)
    for i in range(4):
        k = ctx.timing_collect(i)
        if k.size:
            per_run += k.reshape(reps, -1).sum(axis=1)
    ctx.timing_enable(0)
    return float(np.median(per_run)) * 1e3


def test_the_planners_choice_is_close_to_the_best_forced_plan():
    ctx = native.Context(0)
    ids = np.arange(N, dtype=np.int32)
    age = synth.uniform_below(2, N, 100, np.int8)
    st = synth.state_codes(3, N)
    seg = native.DeviceSegment(ctx, [
        (native.DENSE_INT, 4, ids.view(np.uint8), N * 4, synth.block_offsets(N, 4)),
        (native.DENSE_STRING, 2, st.reshape(-1), N * 2, synth.block_offsets(N, 2)),
        (native.DENSE_TINYINT, 1, age.view(np.uint8), N, synth.block_offsets(N, 1))])
    codes = [bytes(c) for c in np.unique(st[:100_000], axis=0)]
    cells = []
    for pct in (3, 30, 99):
        k, t = float(pct), float(int(N * (1.0 - pct / 100.0)))
        cells += [(f"age<{pct} -> age", [2], [(0, LT, k)], [0]),
                  (f"age<{pct} -> id, age", [2, 0], [(0, LT, k)], [1, 0]),
                  (f"age<{pct} and id>=0 -> id, age", [2, 0], [(0, LT, k), (1, GT, -1.0)], [1, 0]),
                  (f"id>{100 - pct}% -> id", [0], [(0, GT, t)], [0]),
                  (f"id>{100 - pct}% -> id, age", [0, 2], [(0, GT, t)], [0, 1])]
    cells += [("state in 2 -> id, state, age", [1, 0, 2], [(0, MATCH, codes[:2])], [1, 0, 2]),
              ("state in 8 -> state", [1], [(0, MATCH, codes[:8])], [0])]
    worst = []
    try:
        for name, used, sels, proj in cells:
            t = {}
            for attempt in range(2):            # (two queries per variant, the faster one counts: HIP-event noise on 20-60 us of kernels)
                for v in (0, 6, 3, 8, 9):
                    ctx.set_tuning(v, 0)
                    q = native.DeviceQuery(ctx, seg, used, sels, proj, 0, 1024)
                    q.run()
                    q.count()                   # the host has seen the count: the plan may adapt once
                    for _ in range(2):
                        q.run()
                    t[v] = min(t.get(v, 1e9), kernels_us(ctx, q))
                    q.close()
                ctx.set_tuning(0, 0)
                if t[0] <= 1.10 * min(t.values()):
                    break
            ratio = t[0] / min(t.values())
            worst.append((ratio, name, t))
            assert ratio <= 1.30, (name, t)
    finally:
        ctx.set_tuning(0, 0)
        seg.close()
        ctx.close()
    ratios = np.array([r for r, _, _ in worst])
    assert (ratios <= 1.10).mean() >= 0.75, sorted(worst, reverse=True)[:5]
