"""GPU suite: the FAILURE paths of the single-pass projection kernel (csrc/imm3_project.hip, k_filter_project), run on the device.

The kernel's work-groups wait on each other's descriptors; a wait that does not resolve (`abandoned`) or a device that another
launch of the kernel owns (`busy`) makes a launch give up on its ROWS only -- every streamer goes on in count + bitmap mode, so
the count (what the RCCL count all-reduce of config C5 sends, Engine.scala:190-196) and the bitmap stay exact, a status flag
tagged with the run's epoch is raised, and the host getters gather the rows from the bitmap (ProjectIterator.next's walk,
Project.scala:37-64, done by the offsets scan + k_gather instead).  Round 3 only simulated this from the host; here:

  * the tools' build of the library (lib/libimm3_ablate.so, `make -C immutable3_amd/csrc ablate`) carries a fault-injection launch
    argument -- work-group k never announces its span j, look-back waits give up after a small poll cap -- and the test shows
    that every work-group drains, the flag is set, the device count word and the all-reduced count are right BEFORE any host
    getter has looked at the run, and the rows come back through the bitmap path;
  * a foreign ticket in the device's lock word (any build) makes every launch find the device busy: same checks, and the query
    keeps the one-launch plan for later runs;
  * graph replays: a replay that goes wrong after a getter has already verified an earlier run is noticed (round 3 kept the stale
    `verified` flag), and two graphs replayed from two contexts without any host ordering give right rows whichever of them the
    device lock refused.

One process loads one build of the library, so the checks run in a child process with IMM3_LIB_PATH set."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABLATE = os.path.join(ROOT, "immutable3_amd", "lib", "libimm3_ablate.so")

WORKER = r'''
import ctypes as C
import sys, time
import numpy as np
import torch  # noqa: F401  (its HIP runtime first: conftest.py says why)
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from conftest import DENSE_INT, DENSE_TINYINT, GT, LT, RawColumn, blocks_of
from immutable3_amd import native, synth

ABANDONED, BUSY = 2, 4
hip = C.CDLL("libamdhip64.so")

def dev_words(ptr, n):
    out = np.zeros(n, np.uint64)
    assert hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(ptr), C.c_size_t(8 * n), C.c_int(2)) == 0
    return out

def run_flags(q):
    """(count, flags of the LAST run) straight from the device words, no library getter involved."""
    head = dev_words(q.device_ptr(1), 10)              # {count, emitted, status, ..., run counter at [8]}
    status, epoch_run = int(head[2]), int(head[8]) - 1
    mine = ((status >> 8) & 0xFFFFFF) == (epoch_run & 0xFFFFFF)
    assert int(dev_words(q.device_ptr(4), 1)[0]) == status
    return int(head[0]), (status & 6) if mine else 0

n = 13_100 * 1024 - 333
a = synth.uniform_int30(31, n)
c = synth.uniform_below(33, n, 100, np.int8)
br = blocks_of(n, 1024)
cols = [RawColumn(DENSE_INT, 4, a, br), RawColumn(DENSE_TINYINT, 1, c, br)]
ctx = native.Context(0)
seg = native.DeviceSegment(ctx, [x.native() for x in cols])
keep = (c > 18) & (c < 30) & (a > 1000)
rows = np.flatnonzero(keep)
bitmap = np.packbits(keep, bitorder="little").tobytes()
sels = [(0, GT, 18.0), (0, LT, 30.0), (1, GT, 1000.0)]

def check_rows(q, tag):
    idx, vals = q.fetch_rows()
    assert idx.size == rows.size and (idx == rows).all(), tag
    assert vals[0].tobytes() == np.ascontiguousarray(a[rows]).tobytes() and vals[1].tobytes() == np.ascontiguousarray(c[rows]).tobytes(), tag
    assert q.bitmap().tobytes() == bitmap[: q.total_words * 8].ljust(q.total_words * 8, b"\0"), tag

_ff = np.full(rows.size, 0xFFFFFFFF, np.uint32)
def poison(q):
    """Overwrite the query's row-index array on the device (the caller has synchronised): rows left over from an earlier, good run
    must not pass for the rows of a run that gave up on them."""
    assert hip.hipMemcpy(C.c_void_p(q.device_ptr(2)), C.c_void_p(_ff.ctypes.data), C.c_size_t(4 * rows.size), C.c_int(1)) == 0

def fresh(cx=None, sg=None):
    cx, sg = cx or ctx, sg or seg
    cx.set_tuning(202, 0)          # two tiles per range, fixed at creation: 819 spans = four rounds of spans for one work-group per CU
    try:
        q = native.DeviceQuery(cx, sg, [1, 0], sels, [1, 0], 0)
    finally:
        cx.set_tuning(0, 0)
    assert q.plan()["single_pass"] and q.plan()["P"] == 2, q.plan()
    q.reserve_rows(n)              # the row arrays exist (and can be poisoned) before the first run
    cx.sync()
    poison(q)
    return q

comm = native.Comm(ctx, 1, 0, native.comm_unique_id())    # one-rank RCCL communicator: the real collective

# ---- 0. no fault: the baseline
q = fresh()
q.run(); ctx.sync()
assert run_flags(q) == (rows.size, 0)
check_rows(q, "baseline")
assert q.plan()["ran_single_pass"] and q.plan()["abandoned_runs"] == 0 and q.plan()["busy_runs"] == 0
spans, grid = q.plan()["spans"], q.plan()["grid"]
assert spans > 2 * grid, (spans, grid)                 # several rounds of spans: the fault sits in the second one
q.close()

# ---- 1. a span that is never announced: the round never completes, every waiter runs into the poll cap
for wg, span in ((5, 1), (0, 0), (grid - 1, 2)):
    ctx.inject_fault(wg, span, 3000)
    q = fresh()
    t0 = time.time()
    q.run(); ctx.sync()                                 # every work-group drains: the launch ENDS (and soon)
    dt = time.time() - t0
    assert dt < 5.0, dt
    cnt, flags = run_flags(q)
    assert cnt == rows.size, (wg, span, cnt, rows.size)                 # the device word the count all-reduce sends is right ...
    assert flags & ABANDONED and not flags & BUSY, (wg, span, flags)    # ... and the run is flagged, for this epoch
    assert comm.allreduce_count([q]) == rows.size                       # through RCCL, before any getter has settled the run
    ctx.inject_fault(-1, -1, 0)
    assert q.count() == rows.size
    check_rows(q, ("abandoned", wg, span))                              # rows through the bitmap path
    p = q.plan()
    assert p["abandoned_runs"] == 1 and not p["single_pass"] and not p["ran_single_pass"], p
    q.run()                                                             # the query keeps the bitmap path from now on
    check_rows(q, ("after abandoned", wg, span))
    q.close()
    print("abandoned ok", wg, span, "launch + sync %.3f s" % dt, flush=True)

# ---- 2. the device is busy: a foreign ticket in the lock word
q = fresh()
prev = ctx.debug_device_lock(0xDEAD0001)
assert prev == 0, hex(prev)                                             # (every earlier launch handed the device back)
q.run(); ctx.sync()
cnt, flags = run_flags(q)
assert cnt == rows.size and flags & BUSY, (cnt, flags)
assert comm.allreduce_count([q]) == rows.size
check_rows(q, "busy")
p = q.plan()
assert p["busy_runs"] == 1 and p["abandoned_runs"] == 0 and p["single_pass"], p      # this run only
assert ctx.debug_device_lock(0) == 0xDEAD0001                           # (a refused launch never touches the owner's ticket)
poison(q)
q.run(); ctx.sync()
assert run_flags(q) == (rows.size, 0)                                   # the earlier run's flags are not this run's
check_rows(q, "after busy")
assert q.plan()["ran_single_pass"] and q.plan()["busy_runs"] == 1
print("busy ok", flush=True)

# ---- 3. a replay that goes wrong AFTER a getter has verified an earlier run of the same query
with ctx.capture() as cap:
    q.run()
cap.graph.launch()
check_rows(q, "replay")                                                 # sets "verified"
ctx.debug_device_lock(0xDEAD0003)
poison(q)
cap.graph.launch(); ctx.sync()
cnt, flags = run_flags(q)
assert cnt == rows.size and flags & BUSY
check_rows(q, "busy replay")                                            # round 3: stale `verified` -> incomplete rows returned
assert q.plan()["busy_runs"] == 2, q.plan()
ctx.debug_device_lock(0)
poison(q)
cap.graph.launch()
check_rows(q, "replay after busy replay")
assert q.plan()["ran_single_pass"], q.plan()
cap.graph.close()
q.close()
print("stale-verified ok", flush=True)

# ---- 4. two graphs replayed from two contexts, nothing orders them on the host
ctx2 = native.Context(0)
seg2 = native.DeviceSegment(ctx2, [x.native() for x in cols])
qa = fresh()
qb = fresh(ctx2, seg2)
for qq in (qa, qb):
    qq.run()
ctx.sync(); ctx2.sync()
with ctx.capture() as ca:
    qa.run()
with ctx2.capture() as cb:
    qb.run()
busy = 0
for rnd in range(12):
    poison(qa); poison(qb)
    for _ in range(3):                                                  # both streams get a few launches deep
        ca.graph.launch(); cb.graph.launch()
    ctx.sync(); ctx2.sync()
    for qq in (qa, qb):
        cnt, flags = run_flags(qq)
        assert cnt == rows.size and not flags & ABANDONED, (rnd, cnt, flags)   # a refused launch never makes the other one time out
        busy += bool(flags & BUSY)
        check_rows(qq, ("two graphs", rnd))
        assert qq.plan()["single_pass"] and qq.plan()["abandoned_runs"] == 0, qq.plan()
print("two graphs ok; launches that found the device busy:", busy, "of", 24, flush=True)
ca.graph.close(); cb.graph.close()
qa.close(); qb.close()
seg2.close(); ctx2.close()

# ---- 5. a TABLE query (one launch over three segments, csrc/imm3_project_table.hip): the same failure paths -- a span that is never
# announced, a busy device -- leave count and bitmap exact, and the rows come back through the table's bitmap path
cuts = [0, 4_000 * 1024 + 1, 4_000 * 1024 + 1 + 5_000 * 1024 + 777, n]
tsegs = []
for lo_, hi_ in zip(cuts[:-1], cuts[1:]):
    m = hi_ - lo_
    tsegs.append(native.DeviceSegment(ctx, [RawColumn(DENSE_INT, 4, a[lo_:hi_], blocks_of(m, 1024)).native(), RawColumn(DENSE_TINYINT, 1, c[lo_:hi_], blocks_of(m, 1024)).native()]))
table = native.DeviceTable(ctx, tsegs)
starts = np.array(cuts[:-1], dtype=np.int64)

def check_table_rows(q, tag):
    idx, vals = q.fetch_rows()
    seg_of, row_of = q.locate_rows(idx)
    assert idx.size == rows.size and (starts[seg_of] + row_of == rows).all(), tag
    assert vals[0].tobytes() == np.ascontiguousarray(a[rows]).tobytes() and vals[1].tobytes() == np.ascontiguousarray(c[rows]).tobytes(), tag
    words = q.bitmap()
    fb, fw = q.segment_starts()
    for si in range(3):
        want = np.packbits(keep[cuts[si]: cuts[si + 1]], bitorder="little")
        want = np.concatenate([want, np.zeros((-want.size) % 8, np.uint8)]).view("<u8")
        assert words[int(fw[si]): int(fw[si]) + want.size].tolist() == want.tolist(), (tag, si)

def fresh_table():
    ctx.set_tuning(202, 0)
    try:
        q = native.DeviceQuery(ctx, table, [1, 0], sels, [1, 0], 0, 1024)
    finally:
        ctx.set_tuning(0, 0)
    assert q.plan()["single_pass"] and q.plan()["P"] == 2, q.plan()
    q.reserve_rows(n)
    ctx.sync()
    poison(q)
    return q

q = fresh_table()
q.run(); ctx.sync()
assert run_flags(q) == (rows.size, 0)
check_table_rows(q, "table baseline")
assert q.plan()["ran_single_pass"]
q.close()
for wg, span in ((7, 1), (0, 0)):
    ctx.inject_fault(wg, span, 3000)
    q = fresh_table()
    q.run(); ctx.sync()
    cnt, flags = run_flags(q)
    assert cnt == rows.size and flags & ABANDONED and not flags & BUSY, (wg, span, cnt, flags)
    assert comm.allreduce_count([q]) == rows.size
    ctx.inject_fault(-1, -1, 0)
    check_table_rows(q, ("table abandoned", wg, span))
    p = q.plan()
    assert p["abandoned_runs"] == 1 and not p["single_pass"] and not p["ran_single_pass"], p
    q.run()
    check_table_rows(q, ("table after abandoned", wg, span))
    q.close()
q = fresh_table()
assert ctx.debug_device_lock(0xDEAD0005) == 0
q.run(); ctx.sync()
cnt, flags = run_flags(q)
assert cnt == rows.size and flags & BUSY, (cnt, flags)
check_table_rows(q, "table busy")
assert q.plan()["busy_runs"] == 1 and q.plan()["single_pass"], q.plan()
assert ctx.debug_device_lock(0) == 0xDEAD0005
poison(q)
q.run(); ctx.sync()
assert run_flags(q) == (rows.size, 0)
check_table_rows(q, "table after busy")
assert q.plan()["ran_single_pass"]
q.close()
table.close()
for t_ in tsegs:
    t_.close()
print("table ok", flush=True)

# ---- 6. a co-resident kernel on the communicator's stream (the G > 1 shape: one launch per pass + one collective per pass).  The
# stand-in has the footprint of RCCL's all-reduce kernel and cannot share a CU with a work-group of the one-launch projection; while a
# communicator is attached the projection leaves one CU per XCD free for it (imm3_api.cpp: single_pass_run_grid).  Passes under
# it: exact counts on the device, exact rows, no abandoned and no busy run -- with the reservation and (tuning 16) without it.
import torch
PASSES = 40
dlog = torch.zeros(PASSES + 4, dtype=torch.int64, device="cuda")
cus = torch.cuda.get_device_properties(0).multi_processor_count
for variant, want_grid in ((0, cus - 8), (16, cus)):
    ctx.set_tuning(variant, 0)
    try:
        comm.debug_standin(1, 20)      # (a communicator whose collectives launch kernels: a one-rank RCCL all-reduce launches none)
        ctx.set_tuning(204 if variant == 0 else 16, 0)      # (P = 4: more spans than CUs, so the grid is the reservation's to decide)
        q = native.DeviceQuery(ctx, seg, [1, 0], sels, [1, 0], 0)
        ctx.set_tuning(variant, 0)
        q.run(); assert q.count() == rows.size
        q.reserve_rows(rows.size + 1024)
        q.run(); ctx.sync()
        assert q.plan()["single_pass"] and q.plan()["grid"] == min(want_grid, q.plan()["spans"]), (variant, q.plan(), want_grid)
        poison(q)
        for wgs, us in ((1, 20), (4, 60)):
            comm.debug_standin(wgs, us)
            dlog.zero_(); torch.cuda.synchronize()
            for i in range(PASSES):
                q.run()
                comm.allreduce_count([q], device_out=dlog.data_ptr() + 8 * i, wait=False)
            torch.cuda.synchronize()
            assert dlog[:PASSES].tolist() == [rows.size] * PASSES, (variant, wgs, us)
            assert run_flags(q) == (rows.size, 0)
            check_rows(q, ("under the stand-in", variant, wgs, us))
            p = q.plan()
            assert p["ran_single_pass"] and p["abandoned_runs"] == 0 and p["busy_runs"] == 0, p
        comm.debug_standin(0, 0)
        q.close()
    finally:
        ctx.set_tuning(0, 0)
print("co-resident kernel ok", flush=True)
# ---- 7. the small-limit gather (k_limit_gather: offsets scan + gather in one launch): its look-back over the counts of the
# lower-numbered work-groups is bounded; a launch whose wait ran out tags the finish block and a getter gathers with k_scan + k_gather
LIM = 100
for wg in (0, 3):
    q = native.DeviceQuery(ctx, seg, [1, 0], sels, [1, 0], LIM)
    q.run()
    idx, vals = q.fetch_rows()
    assert (idx == rows[:LIM]).all() and q.plan()["limit_gather_gave_up"] == 0, ("limit baseline", q.plan())
    ctx.inject_fault(wg, 0, 200)
    t0 = time.time()
    q.run(); ctx.sync()
    assert time.time() - t0 < 5.0
    idx, vals = q.fetch_rows()
    ctx.inject_fault(-1, -1, 0)
    assert idx.size == LIM and (idx == rows[:LIM]).all(), ("limit gather gave up", wg)
    assert vals[0].tobytes() == np.ascontiguousarray(a[rows[:LIM]]).tobytes() and vals[1].tobytes() == np.ascontiguousarray(c[rows[:LIM]]).tobytes()
    assert q.plan()["limit_gather_gave_up"] == 1, q.plan()
    q.run()
    idx, vals = q.fetch_rows()
    assert (idx == rows[:LIM]).all() and q.plan()["limit_gather_gave_up"] == 1, q.plan()
    q.close()
print("limit gather ok", flush=True)
comm.close(); seg.close(); ctx.close()
print("FAULT-INJECTION-OK", flush=True)
'''


def test_abandoned_and_busy_runs_on_the_device(tmp_path):
    if not os.path.exists(ABLATE):
        from immutable3_amd.build import build_native
        build_native()
    assert os.path.exists(ABLATE), "make -C immutable3_amd/csrc ablate"
    script = tmp_path / "fault_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, IMM3_LIB_PATH=ABLATE)
    r = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=900)
    sys.stdout.write(r.stdout[-4000:])
    sys.stderr.write(r.stderr[-4000:])
    assert r.returncode == 0 and "FAULT-INJECTION-OK" in r.stdout


def test_the_shipped_library_refuses_the_fault_hooks():
    """The fault-injection argument is inert in the shipped kernel and the device-lock hook is not in the shipped library at all
    (round 4 shipped it: any caller could park every one-launch query of a device on its fallback); both say so.  The busy path
    itself is run by the worker above, on the tools' build."""
    import numpy as np
    from conftest import DENSE_INT, DENSE_TINYINT, GT, RawColumn, blocks_of
    from immutable3_amd import native, synth
    ctx = native.Context(0)
    with pytest.raises(native.Imm3Error):
        ctx.inject_fault(3, 0, 100)
    ctx.inject_fault(-1, -1, 0)
    with pytest.raises(native.Imm3Error):
        ctx.debug_device_lock(0xBEEF0001)
    n = 300 * 1024 + 17
    a = synth.uniform_int30(5, n)
    c = synth.uniform_below(6, n, 100, np.int8)
    br = blocks_of(n, 1024)
    seg = native.DeviceSegment(ctx, [RawColumn(DENSE_INT, 4, a, br).native(), RawColumn(DENSE_TINYINT, 1, c, br).native()])
    rows = np.flatnonzero((c > 79) & (a > 5))
    ctx.set_tuning(12, 0)          # (the one launch, whatever the cost model makes of 300 K rows)
    try:
        q = native.DeviceQuery(ctx, seg, [1, 0], [(0, GT, 79.0), (1, GT, 5.0)], [1, 0], 0)
    finally:
        ctx.set_tuning(0, 0)
    assert q.plan()["single_pass"]
    q.run()                        # (the refused hook left the device's ticket word alone)
    idx, vals = q.fetch_rows()
    assert (idx == rows).all() and vals[0].tobytes() == np.ascontiguousarray(a[rows]).tobytes()
    assert q.plan()["ran_single_pass"] and q.plan()["busy_runs"] == 0
    q.close()
    seg.close()
    ctx.close()
