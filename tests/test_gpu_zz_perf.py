"""GPU suite, LAST file of the run (`zz`): every assertion on a TIME lives here, so that under `pytest -x` a noisy box can
only ever fail a timing check -- never hide a parity test behind it (round 4 had them in test_gpu_plan_quality.py and
test_gpu_large.py, both of which sort in front of four files of parity tests).  Timing is taken where it is least noisy:

  * the headline kernel on the device's own 100 MHz clock (first work-group entry to last work-group exit, stamped by the
    kernel: imm3_ctx_devclock_*), no event packets, no host;
  * whole queries (one to three launches each) as ONE hipGraph of ten runs, wall clock around launch + sync, best of three:
    HIP event pairs read ~4 us high per launch and jitter by as much on 20-60 us of kernels (DESIGN finding 35), which is
    what made round 4 loosen the plan-quality bounds; the bounds are back at 1.25 x per cell and 80 % of the cells within 10 %.

North star (BASELINE.json): >= 60 % of the MI355X HBM-read roofline on RangeFilter over a 100 M-row DENSE_INT column.
Plan quality: the plan the library picks (cost model, csrc/imm3_plan.h) against the best plan the
tuning hook can force, on a handful of cells of tools/plan_sweep.py's grid at 16 M rows; the committed sweep
(profiles/r04_plan_sweep.*) is the full grid."""
import time

import numpy as np
import pytest

from immutable3_amd import native, synth

pytestmark = pytest.mark.gpu
GT, LT, MATCH = native.GT, native.LT, native.MATCH


def test_headline_kernel_meets_the_north_star_roofline_target():
    """60 % of 8 TB/s = 85.9 us for the 412.5 MB of algorithmic bytes; the int32 tile kernel has run at 78-86 % on every box."""
    n = 100_000_000
    ctx = native.Context(0)
    v = synth.uniform_int30(1, n)
    seg = native.DeviceSegment(ctx, [(native.DENSE_INT, 4, v.view(np.uint8), n * 4, synth.block_offsets(n, 4))])
    q = native.DeviceQuery(ctx, seg, [0], [(0, GT, float(2 ** 28)), (0, LT, float(3 * 2 ** 28))])
    for _ in range(5):
        q.run_select()
    ctx.sync()
    ctx.devclock_enable(40)
    for _ in range(30):
        q.run_select()
    ctx.sync()
    ms = float(np.median(ctx.devclock_collect()))
    ctx.devclock_enable(0)
    assert q.count() == int(((v > 2 ** 28) & (v < 3 * 2 ** 28)).sum())
    q.close()
    seg.close()
    ctx.close()
    gbps = 4.125 * n / (ms * 1e-3) / 1e9
    assert gbps >= 0.60 * 8000.0, f"scan+select kernel at {gbps:.0f} GB/s ({ms * 1e3:.1f} us, device clock)"


def query_us(ctx, q, runs=10):
    """Microseconds per run of a settled query: ten runs recorded as one hipGraph, best of three launches (wall clock)."""
    ctx.sync()
    with ctx.capture() as cap:
        for _ in range(runs):
            q.run()
    g = cap.graph
    g.launch()
    ctx.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        g.launch()
        ctx.sync()
        best = min(best, time.perf_counter() - t0)
    g.close()
    return best / runs * 1e6


def test_the_planners_choice_is_close_to_the_best_forced_plan():
    N = 16_000_000
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
            for v in (0, 6, 3, 8, 9):
                ctx.set_tuning(v, 0)
                q = native.DeviceQuery(ctx, seg, used, sels, proj, 0, 1024)
                q.run()
                q.count()                       # the host has seen the count: the plan may adapt once
                for _ in range(2):
                    q.run()
                q.row_count()                   # (settled: what the graph records is the query's steady state)
                t[v] = query_us(ctx, q)
                q.close()
            ctx.set_tuning(0, 0)
            ratio = t[0] / min(t.values())
            worst.append((ratio, name, t))
            assert ratio <= 1.25, (name, t)
    finally:
        ctx.set_tuning(0, 0)
        seg.close()
        ctx.close()
    ratios = np.array([r for r, _, _ in worst])
    assert (ratios <= 1.10).mean() >= 0.8, sorted(worst, reverse=True)[:5]
