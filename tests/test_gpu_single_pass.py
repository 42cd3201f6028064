"""GPU suite: the single-pass projection (csrc/imm3_project.hip, k_filter_project) -- ScanOp -> SelectOp* -> ProjectOp of one
uniform segment in ONE launch, which is ProjectIterator.next's one walk over the set bits
(engine/src/main/scala/immutabledb/engine/operator/Project.scala:37-64).  Through the C ABI, bit-exact: bitmap, count,
ascending row order, values.  Covered: every column-kind instance at every fill level with several rounds of spans and a
partial last tile (numpy at that size, the oracle at a smaller one), dense survivors (ranges that outgrow their LDS ring are
unpacked from the source columns; once the host has seen the count, later runs use shorter ranges),
reservations that are too small (the host gathers again from the bitmap), an abandoned run (the host answers through the
bitmap path), graph replays (the descriptors' epochs), a second mention of a column in the SELECT list, and two contexts
launching the kernel at once (the launches are chained: its work-groups wait on each other and need the device)."""
import ctypes as C
import threading

import numpy as np
import pytest

from conftest import DENSE_INT, DENSE_STRING, DENSE_TINYINT, GT, LT, MATCH, RawColumn, blocks_of
from immutable3_amd import native, synth
from test_gpu_parity import check

pytestmark = pytest.mark.gpu

# predicate columns of the instance (indices into [a:i32, b:i32, c:i8, d:i8, s:s2]): the 16 kind combinations of k_filter_project
SHAPES = [[0], [2], [4], [0, 1], [0, 2], [2, 3], [0, 4], [2, 4], [0, 1, 2], [0, 2, 3], [2, 3, 4], [0, 1, 4], [0, 2, 4]]


def pinned_query(ctx, *args):
    """A query on the plan made at creation -- the one launch where the SELECT list allows it -- whatever the cost model predicts for
    its size and survivors (tuning variant 12: the tests of that kernel's own paths)."""
    ctx.set_tuning(12, 0)
    try:
        return native.DeviceQuery(ctx, *args)
    finally:
        ctx.set_tuning(0, 0)


def plan_kind(p):
    return "one launch" if p["single_pass"] else ("records" if p["records"] else "bitmap")


def model_accepts(kind, n, data, used, sels, proj, keep, clustered=False, slack=1.10):
    """Is `kind` a plan the cost model (immutable3_amd/plan_model.py = csrc/imm3_plan.h) could have picked for this query?  Its
    predicted cost must be within `slack` of the cheapest plan's (the library decides on a sampled estimate of the survivors and
    keeps a plan in use unless another is predicted 3 % cheaper, so near ties go either way)."""
    from immutable3_amd import plan_model
    width = lambda u: 2 if data[u].ndim == 2 else data[u].dtype.itemsize
    pred_cols = list(dict.fromkeys(used[i] for i, _, _ in sels))
    pred = [(width(u), max([len(v) for i, op, v in sels if used[i] == u and op == MATCH] + [0])) for u in pred_cols]
    seen, pj = [], []
    for j in proj:
        u = used[j]
        first = u in pred_cols and u not in seen
        if first:
            seen.append(u)
        pj.append((width(u), first))
    n32 = sum(1 for w, _ in pred if w == 4)
    dwords = 1 + n32
    rec_bytes = 4 * (dwords if dwords <= 2 else 4)
    sigma = float(keep.mean())
    sloc, full = (1.0, 1.0) if clustered else (sigma, 0.0)
    letters = {"one launch": "A", "records": "B", "bitmap": "C"}
    cost = {k: plan_model.cost(l, n, sigma, sloc, full, pred, pj, rec_bytes) for k, l in letters.items()}
    eligible = dict(cost)
    gathers = not all(f for _, f in pj)
    if gathers and (not any(w == 4 and not f for w, f in pj) or any(w == 2 for w, _ in pred)):
        eligible.pop("one launch")                      # gathered columns, and none of them int32 (nothing to stream) or a string predicate
    if not any(f for _, f in pj):
        eligible.pop("records")                         # no predicate column projected: records buy nothing
    return kind in eligible and cost[kind] <= slack * min(eligible.values()), cost




@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def big(ctx):
    """13.4 M rows: 13 100 tiles are several rounds of spans for one work-group per CU, and the last tile is partial."""
    n = 13_100 * 1024 - 333
    a = synth.uniform_int30(31, n)
    b = np.arange(n, dtype=np.int32)                      # a sorted key: range predicates on it give CLUSTERED survivors
    c = synth.uniform_below(33, n, 100, np.int8)
    d = (synth.uniform_below(34, n, 100, np.int8) - 50).astype(np.int8)
    s = synth.state_codes(35, n)
    br = blocks_of(n, 1024)
    cols = [RawColumn(DENSE_INT, 4, a, br), RawColumn(DENSE_INT, 4, b, br), RawColumn(DENSE_TINYINT, 1, c, br),
            RawColumn(DENSE_TINYINT, 1, d, br), RawColumn(DENSE_STRING, 2, s, br)]
    seg = native.DeviceSegment(ctx, [x.native() for x in cols])
    yield n, [a, b, c, d, s], seg
    seg.close()


def _predicates(data, pred_cols, level, pos):
    """keep = AND of one predicate per column at a fill level: none / few (~2 %) / some (~10 % each... ) / most / all."""
    n = data[0].shape[0]
    t32 = {"none": 2.0 ** 31, "few": 0.98 * 2 ** 30, "some": 0.9 * 2 ** 30, "most": 0.1 * 2 ** 30, "all": -1.0}
    tb = {"none": float(n), "few": 0.98 * n, "some": 0.9 * n, "most": 0.1 * n, "all": -1.0}           # on the sorted key: clustered
    t8 = {"none": 127.0, "few": 97.0, "some": 89.0, "most": 5.0, "all": -1.0}
    t8d = {"none": 127.0, "few": 47.0, "some": 39.0, "most": -45.0, "all": -128.5}
    codes = {"none": [b"??"], "few": [b"CA"], "some": [b"CA", b"NY", b"TX", b"WA", b"VA"], "most": None, "all": None}
    sels, keep = [], np.ones(n, bool)
    for pc in pred_cols:
        if pc == 4:
            lst = codes[level]
            if lst is None:
                uniq = [bytes(x) for x in np.unique(data[4], axis=0)]
                lst = uniq[:8]                                  # (IN-lists above 8 values go through the generic kernel)
            sels.append((pos[pc], MATCH, lst))
            m = np.zeros(n, bool)
            for v in lst:
                m |= (data[4][:, 0] == v[0]) & (data[4][:, 1] == v[1])
            keep &= m
        else:
            t = {0: t32, 1: tb, 2: t8, 3: t8d}[pc][level]
            sels.append((pos[pc], GT, float(t)))
            keep &= data[pc] > t
    return sels, keep


@pytest.mark.parametrize("pred_cols", SHAPES)
def test_every_instance_at_every_fill_level(ctx, big, pred_cols):
    n, data, seg = big
    used = pred_cols or [0]
    pos = {u: i for i, u in enumerate(used)}
    for level in ("none", "few", "some", "most", "all"):
        sels, keep = _predicates(data, pred_cols, level, pos)
        ctx.set_tuning(12, 0)        # (the plan made at creation stands: this file is about the one launch, which the cost model would leave for few survivors of narrow columns)
        try:
            q = native.DeviceQuery(ctx, seg, used, sels, list(range(len(used))), 0)
        finally:
            ctx.set_tuning(0, 0)
        plan = q.plan()
        folded_empty = level == "none"          # (a threshold no value passes folds to an empty interval: answered by memset, no kernel)
        assert (plan["single_pass"] and not plan["records"]) or folded_empty, (pred_cols, plan)     # the path this file is about
        for _ in range(2):                                                        # twice: rings, descriptors and epochs are reused
            q.run()
        assert q.plan()["ran_single_pass"] or folded_empty
        rows = np.flatnonzero(keep)
        assert q.count() == rows.size, (level, pred_cols)
        assert q.bitmap().tobytes() == np.packbits(keep, bitorder="little").tobytes()[: q.total_words * 8].ljust(q.total_words * 8, b"\0"), (level, pred_cols)
        idx, vals = q.fetch_rows()
        assert q.plan()["ran_single_pass"] or folded_empty, "the run was abandoned and answered through the bitmap path"
        assert idx.size == rows.size and (idx == rows).all(), (level, pred_cols)
        for j, u in enumerate(used):
            got = vals[j]
            assert got.tobytes() == np.ascontiguousarray(data[u][rows]).tobytes(), (level, pred_cols, u)
        q.close()


def test_against_the_oracle_with_ragged_tail_and_second_mentions(ctx, oracle):
    rng = np.random.default_rng(5)
    for n in (1, 1000, 1024, 70_001, 300_000 + 37):
        ids = rng.integers(0, 1000, size=n).astype(np.int32)
        age = rng.integers(0, 100, size=n).astype(np.int8)
        st = np.array([list(x) for x in rng.choice([b"CA", b"NY", b"TX"], size=n)], dtype=np.uint8).reshape(n, 2)
        br = blocks_of(n, 1024)
        cols = [RawColumn(DENSE_INT, 4, ids, br), RawColumn(DENSE_TINYINT, 1, age, br), RawColumn(DENSE_STRING, 2, st, br)]
        sels = [(0, GT, 18.0), (0, LT, 60.0), (1, GT, 100.0), (2, MATCH, [b"CA", b"TX"])]
        check(ctx, oracle, cols, [1, 0, 2], sels, proj=[1, 0, 2])
        check(ctx, oracle, cols, [1, 0, 2], sels, proj=[1, 1, 0, 1])        # second mentions are gathered: not the single-pass plan, same rows
        check(ctx, oracle, cols, [1, 0], sels[:3], proj=[1, 0], reserve=7)   # reservation too small: gathered again from the bitmap at fetch time
        check(ctx, oracle, cols, [1, 0], sels[:3], proj=[1, 0], reserve=n + 5)


def test_plan_only_when_every_select_list_column_is_a_predicate_column(ctx, big):
    n, data, seg = big
    ctx.set_tuning(12, 0)      # (which plans a SELECT list is eligible for: the plan made at creation, before the cost model looks at the survivors)
    try:
        q = native.DeviceQuery(ctx, seg, [2, 0], [(0, GT, 18.0), (0, LT, 30.0), (1, GT, 5.0)], [1, 0], 0)
        assert q.plan()["single_pass"]
        q.close()
        q = native.DeviceQuery(ctx, seg, [4, 0], [(0, MATCH, [b"CA"])], [1, 0], 0)             # id is gathered, the string predicate column projected: records + k_emit
        assert not q.plan()["single_pass"] and q.plan()["records"]
        q.close()
    finally:
        ctx.set_tuning(0, 0)
    q = native.DeviceQuery(ctx, seg, [4, 0], [(0, MATCH, [b"CA"])], [1], 0)                    # no predicate column projected: records would buy nothing
    assert not q.plan()["single_pass"] and not q.plan()["records"]
    q.close()
    q = native.DeviceQuery(ctx, seg, [2, 0], [(0, GT, 18.0), (1, GT, 5.0)], [1, 0], 10)        # limit: the bitmap path
    assert not q.plan()["single_pass"] and not q.plan()["records"]
    q.close()


def test_reservation_too_small_and_abandoned_run_are_answered_from_the_bitmap(ctx, big):
    n, data, seg = big
    a, b, c = data[0], data[1], data[2]
    keep = (c > 89) & (a > 0.5 * 2 ** 30)
    rows = np.flatnonzero(keep)
    sels = [(0, GT, 89.0), (1, GT, float(0.5 * 2 ** 30))]
    q = pinned_query(ctx, seg, [2, 0], sels, [1, 0], 0)
    q.reserve_rows(1000)                          # far too small: the kernel stops writing at the capacity, the count stays exact
    q.run()
    assert q.count() == rows.size
    idx, vals = q.fetch_rows()
    assert (idx == rows).all() and (vals[0].view("<i4").reshape(-1) == a[rows]).all() and (vals[1].view(np.int8).reshape(-1) == c[rows]).all()
    # an abandoned run: what the kernel leaves behind when a prefix never comes (status bit 1, tagged with the run's epoch: the run
    # counter at d_total + 8 words, minus the bump of the run itself).  Simulated by setting the word between the run and the fetch:
    # the host must answer through the bitmap path, and keep that path for this query.  (The kernel's own failure paths run on the
    # device in tests/test_gpu_fault_injection.py.)  A flag with another run's tag is ignored.
    q.run()
    assert q.plan()["ran_single_pass"]
    hip = C.CDLL("libamdhip64.so")
    ctx.sync()
    epoch = np.zeros(1, dtype=np.uint64)
    assert hip.hipMemcpy(C.c_void_p(epoch.ctypes.data), C.c_void_p(q.device_ptr(1) + 64), C.c_size_t(8), C.c_int(2)) == 0
    stale = np.array([(((int(epoch[0]) - 2) & 0xFFFFFF) << 8) | 2], dtype=np.uint64)
    assert hip.hipMemcpy(C.c_void_p(q.device_ptr(4)), C.c_void_p(stale.ctypes.data), C.c_size_t(8), C.c_int(1)) == 0
    idx, vals = q.fetch_rows()
    assert (idx == rows).all() and q.plan()["single_pass"] and q.plan()["ran_single_pass"] and q.plan()["abandoned_runs"] == 0
    q.run()
    ctx.sync()
    assert hip.hipMemcpy(C.c_void_p(epoch.ctypes.data), C.c_void_p(q.device_ptr(1) + 64), C.c_size_t(8), C.c_int(2)) == 0
    status = np.array([(((int(epoch[0]) - 1) & 0xFFFFFF) << 8) | 2], dtype=np.uint64)
    assert hip.hipMemcpy(C.c_void_p(q.device_ptr(4)), C.c_void_p(status.ctypes.data), C.c_size_t(8), C.c_int(1)) == 0   # d_total + 2 words
    idx, vals = q.fetch_rows()
    assert (idx == rows).all() and (vals[0].view("<i4").reshape(-1) == a[rows]).all()
    assert not q.plan()["single_pass"] and not q.plan()["ran_single_pass"] and q.plan()["abandoned_runs"] == 1
    q.run()
    assert q.count() == rows.size and (q.fetch_rows()[0] == rows).all()
    q.close()


def test_graph_replays_and_counts_in_the_log(ctx, big):
    """A recorded single-pass run replayed: every replay is a new epoch of the same descriptors."""
    n, data, seg = big
    c, b = data[2], data[1]
    keep = (c > 18) & (c < 30) & (b > 1000)
    rows = np.flatnonzero(keep)
    q = pinned_query(ctx, seg, [2, 1], [(0, GT, 18.0), (0, LT, 30.0), (1, GT, 1000.0)], [1, 0], 0)
    q.run()                                        # allocates the row arrays (room for every row: no reservation needed)
    with ctx.capture() as cap:
        q.run()
    for _ in range(5):
        cap.graph.launch()
    assert q.count() == rows.size
    idx, vals = q.fetch_rows()
    assert q.plan()["ran_single_pass"] and (idx == rows).all() and (vals[0].view("<i4").reshape(-1) == b[rows]).all()
    cap.graph.close()
    q.close()


def test_two_contexts_launch_the_kernel_at_once(big):
    """The kernel's work-groups wait on each other, so two launches must not share the device: launches of different
    contexts (streams) are chained by the library.  Two threads, a context each, the same segment, 20 runs each."""
    n, data, seg = big
    c, a = data[2], data[0]
    errors = []

    def worker(t):
        try:
            cx = native.Context(0)
            cx.set_tuning(12, 0)             # (the one launch, whatever the cost model makes of 13 M rows)
            lo = 10.0 + 7 * t
            keep = (c > lo) & (c < lo + 12) & (a > 1000)
            rows = np.flatnonzero(keep)
            q = native.DeviceQuery(cx, seg, [2, 0], [(0, GT, lo), (0, LT, lo + 12), (1, GT, 1000.0)], [1, 0], 0)
            for _ in range(20):
                q.run()
            assert q.count() == rows.size
            idx, vals = q.fetch_rows()
            assert (idx == rows).all() and (vals[0].view("<i4").reshape(-1) == a[rows]).all()
            assert q.plan()["single_pass"], "a run was abandoned for good"
            q.close()
            cx.close()
        except BaseException as ex:  # noqa: BLE001
            errors.append((t, repr(ex)))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ths) and not errors, errors


def test_steady_state_projections_never_wait_for_the_device(ctx, big):
    """without a reservation an unlimited projection sizes its arrays once -- never (one launch: room for
    every row) or on its first run (the count) -- and every later imm3_query_run only enqueues."""
    n, data, seg = big
    c, a = data[2], data[0]
    shapes = {"one launch": ([2, 0], [(0, GT, 18.0), (0, LT, 30.0), (1, GT, 5.0)], [1, 0], 0),
              "records": ([2, 0], [(0, GT, 97.0)], [1, 0], 1),
              "streamed": ([2, 0], [(0, GT, 18.0), (0, LT, 30.0)], [1, 0], 0),      # (decided by the sample taken at creation)
              "bitmap": ([2, 0, 4], [(0, GT, 18.0), (2, MATCH, [b"CA", b"NY", b"TX", b"WA", b"VA", b"DC", b"CT", b"AL", b"AK"])], [1, 0], 1)}
    for name, (used, sels, proj, first_run_syncs) in shapes.items():
        q = native.DeviceQuery(ctx, seg, used, sels, proj, 0)
        first_run_syncs = 0 if q.plan()["single_pass"] else 1           # (the one launch has room for every row; the others size their arrays from the first count)
        for _ in range(6):
            q.run()
        assert q.plan()["run_syncs"] == first_run_syncs, (name, q.plan())
        rows = q.row_count()
        assert rows == q.count()
        q.close()


def test_tiles_per_range_follow_the_selectivity(ctx, big):
    """A range whose survivors outgrow the streamer's LDS ring is unpacked from the source columns (unpack_dense); once the host has
    read a run's count, later runs use a P at which the ranges fit again -- unless the survivors are clustered so densely (a
    range predicate on the sorted key) that no useful P would.  Same rows either way; a graph recorded with the first P keeps
    replaying correctly next to direct runs with the new one."""
    n, data, seg = big
    a, b, c = data[0], data[1], data[2]
    cases = {
        "spread, int8":        ([2], [(0, GT, 49.0)], c > 49, "smaller"),
        "spread, int8 + int32": ([2, 0], [(0, GT, 29.0), (1, GT, float(0.2 * 2 ** 30))], (c > 29) & (a > 0.2 * 2 ** 30), "smaller"),
        "nearly all, int8":    ([2], [(0, GT, 0.0)], c > 0, "smaller"),
        "clustered, int32":    ([1], [(0, GT, float(n // 2))], b > n // 2, "same"),
        "sparse":              ([2, 0], [(0, GT, 89.0), (1, GT, float(0.5 * 2 ** 30))], (c > 89) & (a > 0.5 * 2 ** 30), "same"),
    }
    for name, (used, sels, keep, expect) in cases.items():
        rows = np.flatnonzero(keep)
        sampled = native.DeviceQuery(ctx, seg, used, sels, list(range(len(used))), 0)     # the plan from the sample taken at creation ...
        ctx.set_tuning(12, 0)                                                              # ... and the one launch pinned, without a sample: what the first count teaches P
        try:
            q = native.DeviceQuery(ctx, seg, used, sels, list(range(len(used))), 0)
        finally:
            ctx.set_tuning(0, 0)
        p0 = q.plan()
        assert p0["single_pass"], (name, p0)
        q.run()
        g = None
        if name == "spread, int8":                     # recorded with the planned P, before the host has seen a count
            with ctx.capture() as cap:
                q.run()
            g = cap.graph
        assert q.count() == rows.size, name            # the host learns the count (and how many ranges outgrew their ring)
        p1 = q.plan()
        assert p1["ran_single_pass"], (name, p1)
        if expect == "smaller":
            assert 2 <= p1["P"] < p0["P"], (name, p0, p1)
        else:
            assert p1["P"] == p0["P"], (name, p0, p1)
        for rnd in range(3):
            if g is not None and rnd == 1:
                g.launch()                             # the old P, between two runs with the new one
            else:
                q.run()
            assert q.count() == rows.size, (name, rnd)
            idx, vals = q.fetch_rows()
            assert q.plan()["ran_single_pass"], (name, rnd)
            assert idx.size == rows.size and (idx == rows).all(), (name, rnd)
            for j, u in enumerate(used):
                assert vals[j].tobytes() == np.ascontiguousarray(data[u][rows]).tobytes(), (name, rnd, u)
        assert q.plan()["P"] == p1["P"], (name, "P keeps still once it fits")
        ps = sampled.plan()
        ok, cost = model_accepts(plan_kind(ps), n, data, used, sels, list(range(len(used))), keep, clustered=name.startswith("clustered"))
        assert ok, (name, ps, cost)                                                    # (at 13 M rows the cost model takes several of these off the one launch)
        if ps["single_pass"]:
            assert (ps["P"] < p0["P"]) == (expect == "smaller"), (name, p0, ps)       # the sample leads to the same side of the planned P
        sampled.run()
        idx, vals = sampled.fetch_rows()
        assert sampled.plan()["ran_single_pass"] == ps["single_pass"] and idx.size == rows.size and (idx == rows).all(), name
        sampled.close()
        if g is not None:
            g.close()
        q.close()


def test_fully_surviving_ranges_are_copied_whatever_the_first_slot(ctx):
    """unpack_dense's straight copy: a range all of whose rows survive is moved with 16-byte loads and stores.  The output slot of its
    first row is arbitrary (the survivors before it): every residue mod 4, with int32, int8 and 2-byte columns in the SELECT list
    (a narrow column whose first output byte is not dword-aligned takes the general walk), a reservation that ends inside a copied
    range, and the segment's partial last tile at the end of the run of survivors."""
    n = 3_000 * 1024 - 77
    rng = np.random.default_rng(21)
    key = np.arange(n, dtype=np.int32)
    a = rng.integers(0, 2 ** 30, size=n).astype(np.int32)
    c = rng.integers(0, 100, size=n).astype(np.int8)
    codes = [b"CA", b"NY", b"TX"]
    s = np.array([list(x) for x in codes], dtype=np.uint8)[rng.integers(0, 3, size=n)]
    data = [key, a, c, s]
    br = blocks_of(n, 1024)
    cols = [RawColumn(DENSE_INT, 4, key, br), RawColumn(DENSE_INT, 4, a, br), RawColumn(DENSE_TINYINT, 1, c, br), RawColumn(DENSE_STRING, 2, s, br)]
    seg = native.DeviceSegment(ctx, [x.native() for x in cols])
    shapes = [([0], [], None),
              ([0, 2], [(1, GT, -1.0)], None),                                          # + int8, every row passes it
              ([0, 3, 2], [(1, MATCH, codes), (2, GT, -1.0)], None),                    # + 2-byte codes + int8, every row passes both
              ([0, 1], [(1, GT, float(0.001 * 2 ** 30))], a > 0.001 * 2 ** 30)]         # a few holes: ranges with and without them
    for shift in (0, 1, 2, 3, 1024 * 7 + 5):
        thr = n // 3 + shift                                           # key > thr: everything behind row thr, one long run
        for used, extra, extra_keep in shapes:
            sels = [(0, GT, float(thr))] + extra
            keep = key > thr
            if extra_keep is not None:
                keep = keep & extra_keep
            rows = np.flatnonzero(keep)
            for reserve in (0, rows.size // 2 + 3):
                q = pinned_query(ctx, seg, used, sels, list(range(len(used))), 0)
                if reserve:
                    q.reserve_rows(reserve)
                assert q.plan()["single_pass"], (used, q.plan())
                for _ in range(2):
                    q.run()
                assert q.count() == rows.size, (shift, used, reserve)
                idx, vals = q.fetch_rows()
                assert idx.size == rows.size and (idx == rows).all(), (shift, used, reserve)
                for j, u in enumerate(used):
                    assert vals[j].tobytes() == np.ascontiguousarray(data[u][rows]).tobytes(), (shift, used, reserve, u)
                q.close()
    seg.close()


def test_gathered_columns_in_the_one_launch_when_the_tuning_hook_forces_it(big):
    """The planner keeps gathered SELECT-list columns out of the single launch (the three launches are faster for them); the
    diagnostics hook (imm3_ctx_set_tuning, variant 8) forces them in for A/B runs, so that code is reachable and has to be
    right: columns gathered by the writers from records (gather_range) and, for ranges that outgrew their ring, from the
    bitmap lines (unpack_dense's second walk).  Variant 6 -- never the single launch -- gives the same rows."""
    n, data, seg0 = big
    a, b, c, d, s2 = data
    ctx8 = native.Context(0)
    br = blocks_of(n, 1024)
    cols = [RawColumn(DENSE_INT, 4, a, br), RawColumn(DENSE_INT, 4, b, br), RawColumn(DENSE_TINYINT, 1, c, br),
            RawColumn(DENSE_TINYINT, 1, d, br), RawColumn(DENSE_STRING, 2, s2, br)]
    seg = native.DeviceSegment(ctx8, [x.native() for x in cols])
    cases = {
        "sparse, one gathered":  ([2, 0], [(0, GT, 89.0)], [0, 1], c > 89),
        "dense, two gathered":   ([2, 0, 4], [(0, GT, 9.0)], [1, 0, 2], c > 9),
        "clustered, gathered narrow + second mention": ([1, 3], [(0, GT, float(n // 2))], [0, 1, 0], b > n // 2),
    }
    try:
        for variant, want_single in ((8, True), (6, False)):
            ctx8.set_tuning(variant, 0)
            for name, (used, sels, proj, keep) in cases.items():
                rows = np.flatnonzero(keep)
                q = native.DeviceQuery(ctx8, seg, used, sels, proj, 0)
                assert q.plan()["single_pass"] == want_single, (variant, name, q.plan())
                for rnd in range(2):                           # (the second run: P adapted to the count the first one showed)
                    q.run()
                    assert q.count() == rows.size, (variant, name, rnd)
                    idx, vals = q.fetch_rows()
                    assert q.plan()["ran_single_pass"] == want_single, (variant, name, rnd)
                    assert idx.size == rows.size and (idx == rows).all(), (variant, name, rnd)
                    for j, pj in enumerate(proj):
                        assert vals[j].tobytes() == np.ascontiguousarray(data[used[pj]][rows]).tobytes(), (variant, name, rnd, j)
                q.close()
    finally:
        ctx8.set_tuning(0, 0)
        seg.close()
        ctx8.close()


def test_gathered_int32_columns_are_streamed_when_the_cost_model_says_so(ctx, big):
    """`select id, age ... where age > 18 and age < 30` (the reference README's example): id is not a predicate column.  Three plans
    exist for such a projection -- survivor records -> offsets scan -> emit (the predicate columns' values ride in the records),
    the same from the bitmap alone (no records: the plain filter kernel, the gather reads every SELECT-list column), and the one
    launch with the gathered int32 columns streamed through it as tile columns that let every value pass.  Which one is the
    fastest depends on the rows, on how many survive and on which predicate columns are projected (csrc/imm3_plan.h); the library
    decides from a sample counted at query creation and, without one (small segments; here: tuning variant 10), from the first run's
    count.  Same rows whatever the plan; tuning variant 9 forces the streamed plan, so its transition runs here whatever the model
    makes of 13 M rows."""
    n, data, seg = big
    a, b, c, d, s2 = data
    cases = {
        # name: used, sels, proj, keep, plan at creation without the sample
        "11 %, id gathered":         ([2, 0], [(0, GT, 18.0), (0, LT, 30.0)], [1, 0], (c > 18) & (c < 30), "records"),
        "10 %, two int32 gathered":  ([2, 0, 1], [(0, GT, 89.0)], [2, 0, 1], c > 89, "records"),
        "second mention stays a gather": ([3, 0], [(0, GT, 30.0)], [1, 0, 1], d > 30, "records"),
        "2 %, int8 predicate projected": ([2, 0], [(0, GT, 97.0)], [1, 0], c > 97, "records"),
        "2 %, string predicate projected": ([4, 0], [(0, MATCH, [b"CA"])], [1, 0], None, "records"),
        "string predicate, not projected": ([4, 0], [(0, MATCH, [b"CA", b"NY", b"TX", b"WA", b"VA"])], [1], None, "bitmap"),
        "only a 1-byte column gathered": ([0, 2], [(0, GT, float(0.8 * 2 ** 30))], [1], a > 0.8 * 2 ** 30, "bitmap"),
        "20 %, no predicate column projected": ([2, 0], [(0, GT, 79.0)], [1], c > 79, "bitmap"),
        "5 %, no predicate column projected": ([2, 0], [(0, GT, 94.0)], [1], c > 94, "bitmap"),
    }

    def check_rows(q, rows, used, proj, tag):
        idx, vals = q.fetch_rows()
        assert idx.size == rows.size and (idx == rows).all(), tag
        for j, pj in enumerate(proj):
            assert vals[j].tobytes() == np.ascontiguousarray(data[used[pj]][rows]).tobytes(), (tag, j)

    for name, (used, sels, proj, keep, want_created) in cases.items():
        if keep is None:
            keep = np.zeros(n, bool)
            for v in sels[0][2]:
                keep |= (s2[:, 0] == v[0]) & (s2[:, 1] == v[1])
        rows = np.flatnonzero(keep)
        sampled = native.DeviceQuery(ctx, seg, used, sels, proj, 0)    # the sample taken at creation decides before the first run ...
        k0 = plan_kind(sampled.plan())
        ok, cost = model_accepts(k0, n, data, used, sels, proj, keep)
        assert ok, (name, sampled.plan(), cost)
        sampled.run()
        check_rows(sampled, rows, used, proj, name)
        ok, cost = model_accepts(plan_kind(sampled.plan()), n, data, used, sels, proj, keep)
        assert ok and sampled.plan()["run_syncs"] == (0 if k0 == "one launch" else 1), (name, sampled.plan(), cost)
        sampled.close()
        ctx.set_tuning(10, 0)                                           # ... without it, the first run's count does
        try:
            q = native.DeviceQuery(ctx, seg, used, sels, proj, 0)
        finally:
            ctx.set_tuning(0, 0)
        assert plan_kind(q.plan()) == want_created, (name, q.plan())
        for rnd in range(3):
            q.run()
            p = q.plan()
            ok, cost = model_accepts(plan_kind(p), n, data, used, sels, proj, keep)
            assert ok and p["ran_single_pass"] == p["single_pass"], (name, rnd, p, cost)
            assert p["run_syncs"] == 1, (name, rnd, p)           # the first run's look at the count, never again
            assert q.count() == rows.size, (name, rnd)
            check_rows(q, rows, used, proj, (name, rnd))
        q.close()
        # forced: the gathered int32 columns are streamed whatever the prediction (where there is one to stream and no string predicate)
        ctx.set_tuning(9, 0)
        try:
            q = native.DeviceQuery(ctx, seg, used, sels, proj, 0)
            for rnd in range(2):
                q.run()
                assert q.count() == rows.size, (name, "forced", rnd)
                check_rows(q, rows, used, proj, (name, "forced", rnd))
            if name in ("11 %, id gathered", "10 %, two int32 gathered", "second mention stays a gather", "20 %, no predicate column projected"):
                assert q.plan()["ran_single_pass"], (name, q.plan())
            q.close()
        finally:
            ctx.set_tuning(0, 0)
    # a reservation stands in for the count the first run would have read: forced to the streamed plan here, it switches before the first run
    ctx.set_tuning(10, 0)
    try:
        q = native.DeviceQuery(ctx, seg, [2, 0], [(0, GT, 18.0), (0, LT, 30.0)], [1, 0], 0)
    finally:
        ctx.set_tuning(0, 0)
    assert not q.plan()["single_pass"] and q.plan()["records"]
    rows = np.flatnonzero((c > 18) & (c < 30))
    ctx.set_tuning(9, 0)
    try:
        q.reserve_rows(rows.size + 100)
    finally:
        ctx.set_tuning(0, 0)
    assert q.plan()["single_pass"] and not q.plan()["records"]
    q.run()
    assert q.plan()["run_syncs"] == 0 and q.plan()["ran_single_pass"]
    idx, vals = q.fetch_rows()
    assert (idx == rows).all() and (vals[0].view("<i4").reshape(-1) == a[rows]).all() and vals[1].tobytes() == c[rows].tobytes()
    q.close()


def test_random_sizes_predicates_and_select_lists(ctx):
    """Differential run against numpy over random segment sizes (one row to a few hundred tiles, mostly not a multiple of the tile),
    random predicates from no survivor to every row, clustered (sorted key) and spread, and random SELECT lists with second
    mentions and columns that are not predicate columns -- three runs each with the count read in between, so the plan the host
    adapts (tiles per range, streamed columns) is exercised at sizes where spans < work-groups and ranges are mostly empty."""
    rng = np.random.default_rng(20261004)
    for case in range(150):
        n = int(rng.choice([1, 63, 1024, 1025, 5000, 70_001, 262_144, 400_000 + int(rng.integers(0, 1024))]))
        # (values one above the types' minima, so that "every row" -- a threshold one below the smallest value -- is still a value of
        # the type: the reference narrows thresholds the JVM way, (byte)(-129.0) = 127, tests/test_gpu_parity.py)
        a = rng.integers(-2 ** 31 + 1, 2 ** 31 - 1, size=n, dtype=np.int64).astype(np.int32)
        b = np.arange(n, dtype=np.int32)
        c = rng.integers(-127, 128, size=n).astype(np.int8)
        d = rng.integers(0, 100, size=n).astype(np.int8)
        data = [a, b, c, d]
        br = blocks_of(n, 1024)
        cols = [RawColumn(DENSE_INT, 4, a, br), RawColumn(DENSE_INT, 4, b, br), RawColumn(DENSE_TINYINT, 1, c, br), RawColumn(DENSE_TINYINT, 1, d, br)]
        seg = native.DeviceSegment(ctx, [x.native() for x in cols])
        used = [int(x) for x in rng.permutation(4)[: int(rng.integers(1, 4))]]
        n_pred = int(rng.integers(1, len(used) + 1))
        sels, keep = [], np.ones(n, bool)
        for i in range(n_pred):
            col = data[used[i]]
            frac = float(rng.choice([0.0, 0.02, 0.1, 0.3, 0.6, 1.0]))
            srt = np.sort(col)
            t = float(srt[min(n - 1, int((1.0 - frac) * n))]) - (1.0 if frac == 1.0 else 0.0) if frac > 0.0 else float(srt[-1])
            sels.append((i, GT, t))
            keep &= col > t
        proj = [int(x) for x in rng.integers(0, len(used), size=int(rng.integers(1, 5)))]
        rows = np.flatnonzero(keep)
        q = native.DeviceQuery(ctx, seg, used, sels, proj, 0)
        for rnd in range(3):
            q.run()
            assert q.count() == rows.size, (case, rnd, n, used, sels, proj, q.plan())
            idx, vals = q.fetch_rows()
            assert idx.size == rows.size and (idx == rows).all(), (case, rnd, n, used, sels, proj, q.plan())
            for j, pj in enumerate(proj):
                assert vals[j].tobytes() == np.ascontiguousarray(data[used[pj]][rows]).tobytes(), (case, rnd, j, n, used, sels, proj, q.plan())
        q.close()
        seg.close()


def test_the_plan_follows_the_cost_model_with_and_without_the_sample(ctx, big):
    """The one-launch kernel costs about the same per row whatever the columns' widths; the plain filter over 1- and 2-byte columns
    is four times cheaper than over an int32 column, and three small launches start cheaper than the one.  `select age ... where
    age > 98` is therefore planned as filter -> offsets scan -> gather from the bitmap when the sample taken at creation shows few
    survivors, a projected string column with few survivors is staged in records, and so on: whatever csrc/imm3_plan.h predicts
    cheapest.  Without a sample (tuning variant 10) the query starts on the one launch and the first count brings it to a plan the
    model accepts.  Same rows whatever the plan."""
    n, data, seg = big
    a, b, c, d, s2 = data
    cases = {
        "int8, 10 %":        ([2], [(0, GT, 89.0)], c > 89, False),
        "int8, 50 %":        ([2], [(0, GT, 49.0)], c > 49, False),
        "int8 + int8, 5 %":  ([2, 3], [(0, GT, 89.0), (1, GT, 0.0)], (c > 89) & (d > 0), False),
        "string, 2 %":       ([4], [(0, MATCH, [b"CA"])], (s2[:, 0] == ord("C")) & (s2[:, 1] == ord("A")), False),
        "string, 10 %":      ([4], [(0, MATCH, [b"CA", b"NY", b"TX", b"WA", b"VA"])], None, False),
        "int8 + int32, 10 %": ([2, 0], [(0, GT, 89.0), (1, GT, -1.0)], (c > 89) & (a > -1), False),
        "int8 + int32, 99 %": ([2, 0], [(0, GT, 0.0), (1, GT, -1.0)], (c > 0) & (a > -1), False),
        "sorted key, 40 %":  ([1], [(0, GT, float(0.6 * n))], b > 0.6 * n, True),
    }
    for name, (used, sels, keep, clustered) in cases.items():
        if keep is None:
            keep = np.zeros(n, bool)
            for v in sels[0][2]:
                keep |= (s2[:, 0] == v[0]) & (s2[:, 1] == v[1])
        rows = np.flatnonzero(keep)
        proj = list(range(len(used)))
        q = native.DeviceQuery(ctx, seg, used, sels, proj, 0)
        ok, cost = model_accepts(plan_kind(q.plan()), n, data, used, sels, proj, keep, clustered)
        assert ok, (name, q.plan(), cost)
        ctx.set_tuning(10, 0)
        try:
            late = native.DeviceQuery(ctx, seg, used, sels, proj, 0)
        finally:
            ctx.set_tuning(0, 0)
        assert plan_kind(late.plan()) == "one launch", (name, late.plan())
        for qq in (q, late):
            for rnd in range(3):
                qq.run()
                assert qq.count() == rows.size, (name, rnd)
                idx, vals = qq.fetch_rows()
                assert idx.size == rows.size and (idx == rows).all(), (name, rnd)
                for j, u in enumerate(used):
                    assert vals[j].tobytes() == np.ascontiguousarray(data[u][rows]).tobytes(), (name, rnd, j)
            ok, cost = model_accepts(plan_kind(qq.plan()), n, data, used, sels, proj, keep, clustered)
            assert ok, (name, qq.plan(), cost)                         # (the first count brought `late` to a plan the model accepts)
            qq.close()


def test_survivors_between_the_sample_points(ctx, big):
    """The sample at creation looks at eight chunks of 64 tiles; a range of the sorted key that lies between two of them shows it no
    survivor at all.  The answer is exact all the same, and the first count puts the query on a plan the cost model accepts for the
    survivors that are really there (1300 fully surviving tiles: 10 % of the rows, in one run)."""
    n, data, seg = big
    a, b, c, d, s2 = data
    n_full = n // 1024
    centres = [(2 * i + 1) * n_full // 16 for i in range(8)]          # the sample's chunks: 64 tiles around each (imm3_api.cpp: sample_tile_ptrs)
    lo_tile, hi_tile = centres[0] + 100, centres[1] - 100
    assert hi_tile - lo_tile > 1000
    lo, hi = lo_tile * 1024 + 7, hi_tile * 1024 - 9
    keep = (b > lo) & (b < hi)
    rows = np.flatnonzero(keep)
    for used, extra, proj in (([1], [], [0]), ([1, 2], [], [0, 1]), ([1, 0], [(1, GT, -1.0)], [1, 0])):
        sels = [(0, GT, float(lo)), (0, LT, float(hi))] + extra
        q = native.DeviceQuery(ctx, seg, used, sels, proj, 0)
        first = plan_kind(q.plan())
        for rnd in range(3):
            q.run()
            assert q.count() == rows.size, (used, rnd)
            idx, vals = q.fetch_rows()
            assert idx.size == rows.size and (idx == rows).all(), (used, rnd)
            for j, pj in enumerate(proj):
                assert vals[j].tobytes() == np.ascontiguousarray(data[used[pj]][rows]).tobytes(), (used, rnd, j)
        ok, cost = model_accepts(plan_kind(q.plan()), n, data, used, sels, proj, keep, clustered=True, slack=1.25)
        assert ok, (used, first, q.plan(), cost)
        q.close()


def test_string_records_sparse_and_dense_tiles(big):
    """Survivor records of a lone 2-byte string column (C4's shape: state in (...) -> id, state, ...) from 2 % to 16 % survivors
    (one, three, five and eight IN-list values), with one and with two gathered columns and a second mention: rows, values and their
    order against numpy; the records plan forced (tuning variant 6: the cost model would take the bitmap path at this size)."""
    n, data, seg0 = big
    a, b, c, d, s2 = data
    ctx6 = native.Context(0)
    br = blocks_of(n, 1024)
    cols = [RawColumn(DENSE_INT, 4, a, br), RawColumn(DENSE_INT, 4, b, br), RawColumn(DENSE_TINYINT, 1, c, br),
            RawColumn(DENSE_TINYINT, 1, d, br), RawColumn(DENSE_STRING, 2, s2, br)]
    seg = native.DeviceSegment(ctx6, [x.native() for x in cols])
    uniq = [bytes(x) for x in np.unique(s2, axis=0)]
    try:
        ctx6.set_tuning(6, 0)
        for m in (1, 3, 5, 8):
            lst = uniq[3:3 + m]
            keep = np.zeros(n, bool)
            for v in lst:
                keep |= (s2[:, 0] == v[0]) & (s2[:, 1] == v[1])
            rows = np.flatnonzero(keep)
            for used, proj in (([4, 1], [1, 0]), ([4, 0, 2], [0, 2, 1, 0])):
                q = native.DeviceQuery(ctx6, seg, used, [(0, MATCH, lst)], proj, 0)
                assert q.plan()["records"] and not q.plan()["single_pass"], q.plan()
                for rnd in range(2):
                    q.run()
                    assert q.count() == rows.size, (m, rnd)
                    assert q.bitmap().tobytes() == np.packbits(keep, bitorder="little").tobytes()[: q.total_words * 8].ljust(q.total_words * 8, b"\0"), (m, rnd)
                    idx, vals = q.fetch_rows()
                    assert idx.size == rows.size and (idx == rows).all(), (m, rnd)
                    for j, pj in enumerate(proj):
                        assert vals[j].tobytes() == np.ascontiguousarray(data[used[pj]][rows]).tobytes(), (m, rnd, j)
                assert q.plan()["records"], q.plan()
                q.close()
    finally:
        ctx6.set_tuning(0, 0)
        seg.close()
        ctx6.close()
