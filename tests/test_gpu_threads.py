"""GPU suite: the threading contract of the C ABI (include/imm3.h, "Threading"; csrc/imm3_sync.h).

The reference calls the path from a FixedThreadPool(cpuCount) with one PipelineThread per segment
(engine/src/main/scala/immutabledb/engine/Engine.scala:176-180,247-262; SqlCli.scala:64), and the Scala drop-in
(integration/scala/immutabledb/operator/GpuOps.scala) gives every PipelineThread of a device the same context.  So: 8 host
threads (ctypes drops the GIL around every call) each create, run, fetch and destroy their own queries -- select-only,
staged projection, limited projection, aggregation, projections of lazily decoded PFOR_INT / snappy columns -- against one
shared set of segments, bit-exact against the oracle, in both shapes the contract allows: ONE context shared by all
threads, and one context per thread.  A graph capture by one thread while the others keep calling is part of it."""
import threading

import numpy as np
import pytest

from conftest import DENSE_INT, DENSE_STRING, DENSE_TINYINT, GT, LT, MATCH, PforColumn, RawColumn, SnappyColumn, blocks_of
from immutable3_amd import native
from oracle import oracle_c, oracle_np

pytestmark = pytest.mark.gpu
N_THREADS = 8
CODES = [b"CA", b"NY", b"TX", b"WA", b"VA", b"DC", b"CT"]
KIND = {"count": native.AGG_COUNT, "min": native.AGG_MIN, "max": native.AGG_MAX}


def _table(n, seed):
    rng = np.random.default_rng(seed)
    br = blocks_of(n, 1024)
    ids = np.arange(n, dtype=np.int32) * 3 + 7
    age = rng.integers(0, 100, size=n).astype(np.int8)
    st = np.array([list(CODES[i]) for i in rng.integers(0, len(CODES), size=n)], dtype=np.uint8).reshape(n, 2)
    val = rng.integers(-10 ** 6, 10 ** 6, size=n).astype(np.int32)
    return [RawColumn(DENSE_INT, 4, ids, br), RawColumn(DENSE_TINYINT, 1, age, br), RawColumn(DENSE_STRING, 2, st, br),
            PforColumn(ids, br), SnappyColumn(DENSE_INT, 4, val, br)]


# (used, sels, proj, limit) -- indices into _table()'s columns
def _shapes(n):
    return [
        ("select", [1, 0], [(0, GT, 18.0), (0, LT, 30.0), (1, GT, 1000.0)], [], 0),
        ("staged", [1, 0], [(0, GT, 18.0), (0, LT, 30.0), (1, GT, 1000.0), (1, LT, float(3 * n - 999))], [1, 0], 0),
        ("staged_gather", [2, 0, 1], [(0, MATCH, [b"CA"])], [1, 0, 2], 0),
        ("limit", [1, 0], [(0, GT, 50.0)], [1], 37),
        ("pfor_project", [1, 3], [(0, LT, 5.0), (1, GT, 30000.0)], [1, 0], 0),       # projects the PFOR_INT column: lazy decode
        ("snappy_project", [4, 1], [(0, GT, 0.0), (1, GT, 90.0)], [0], 0),           # snappy column: lazy decode + CRC table
    ]


AGGS = [("agg_state", [2, 0, 1], [(2, GT, 18.0), (2, LT, 60.0)], [0], [("count", 1), ("max", 2)]),
        ("agg_age", [1, 0], [], [0], [("count", 1), ("min", 1)])]


def _expected(cols, n):
    exp = {}
    for name, used, sels, proj, limit in _shapes(n):
        ucols = [cols[i] for i in used]
        ow, oc = oracle_c.scan_select([c.ocol() for c in ucols], sels, 1024, 1)
        e = {"words": ow, "count": oc}
        if proj:
            size, _, _, _ = oracle_c.layout(ucols[0].ocol(), 1024)
            k, batch, pos, vals, _ = oracle_c.project([c.ocol() for c in ucols], list(proj), limit, 1024, ow)
            starts = np.concatenate([[0], np.cumsum(size.astype(np.int64))])
            e["rows"] = (starts[batch] + pos).astype(np.uint32)
            e["vals"] = [v.tobytes() for v in vals]
        exp[name] = e
    for name, used, sels, group, aggs in AGGS:
        ucols = [cols[i] for i in used]
        words, _ = oracle_c.scan_select([c.ocol() for c in ucols], sels, 1024)
        exp[name] = oracle_c.project_agg([c.ocol() for c in ucols], group, aggs, words)
    return exp


def _decode_groups(cols, used, group, aggs, keys, counts, vals):
    ucols = [cols[i] for i in used]
    got = {}
    for g in range(keys.shape[0]):
        raw = int(keys[g]).to_bytes(8, "little")
        parts, off = [], 0
        for gi in group:
            c = ucols[gi]
            chunk = raw[off: off + c.width]
            parts.append(chunk.decode() if getattr(c, "_dense", c).codec == DENSE_STRING else str(int.from_bytes(chunk, "little", signed=True)))
            off += c.width
        st = []
        for j, (kind, ci) in enumerate(aggs):
            st.append(int(counts[g]) if kind == "count" else int(vals[g, j]))
        got["_".join(parts)] = st
    return got


def _worker(tid, ctx, seg, cols, n, exp, rounds, errors, barrier, capture_ctx=None):
    try:
        barrier.wait(timeout=120)
        shapes = _shapes(n)
        for r in range(rounds):
            order = list(range(len(shapes)))
            order = order[(tid + r) % len(order):] + order[:(tid + r) % len(order)]   # every thread a different interleaving
            for k in order:
                name, used, sels, proj, limit = shapes[k]
                q = native.DeviceQuery(ctx, seg, used, sels, proj, limit, 1024)
                if proj and not limit and (tid + r) % 2:
                    q.reserve_rows(n)                       # (both the reserved and the count-sized projection)
                q.run()
                e = exp[name]
                assert q.count() == e["count"], (name, tid)
                assert q.bitmap().tolist() == e["words"].tolist(), (name, tid)
                if proj:
                    idx, vals = q.fetch_rows()
                    assert idx.tolist() == e["rows"].tolist(), (name, tid)
                    assert [v.tobytes() for v in vals] == e["vals"], (name, tid)
                q.close()
            for name, used, sels, group, aggs in AGGS:
                q = native.DeviceQuery(ctx, seg, used, sels, (), 0, 1024, group_cols=group, aggs=[(KIND[a], c) for a, c in aggs])
                q.run()
                keys, first, counts, vals = q.fetch_groups()
                got = _decode_groups(cols, used, group, aggs, keys, counts, vals)
                want = {k: [int(float(x)) if not isinstance(x, int) else x for x in v] for k, v in exp[name].items()}
                assert sorted(got) == sorted(want), (name, tid)
                for key in got:
                    assert got[key] == want[key], (name, tid, key)
                q.close()
            if capture_ctx is not None and tid == 0:
                # one thread records a graph on the SHARED context while the others keep calling: their calls wait at the
                # context's gate, nothing of theirs lands in the graph
                name, used, sels, proj, limit = shapes[1]
                q = native.DeviceQuery(capture_ctx, seg, used, sels, proj, limit, 1024)
                q.reserve_rows(n)
                q.run()
                with capture_ctx.capture() as cap:
                    q.run()
                for _ in range(3):
                    cap.graph.launch()
                idx, vals = q.fetch_rows()
                assert idx.tolist() == exp[name]["rows"].tolist() and [v.tobytes() for v in vals] == exp[name]["vals"]
                cap.graph.close()
                q.close()
    except BaseException as ex:  # noqa: BLE001 -- reported by the main thread
        errors.append((tid, repr(ex)))


def _agg_expected_as_ints(exp):
    return exp


@pytest.fixture(scope="module")
def table():
    n = 150_000 + 321
    cols = _table(n, 77)
    return n, cols, _expected(cols, n)


def _run_threads(target_args):
    errors = []
    barrier = threading.Barrier(N_THREADS)
    ths = [threading.Thread(target=_worker, args=(t, *target_args(t), errors, barrier)) for t in range(N_THREADS)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in ths), "a worker thread hangs"
    assert not errors, errors


def test_eight_threads_share_one_context(table):
    n, cols, exp = table
    ctx = native.Context(0)
    seg = native.DeviceSegment(ctx, [c.native() for c in cols])   # fresh: the lazy decodes are raced by the threads
    _run_threads(lambda t: (ctx, seg, cols, n, exp, 3))
    seg.close()
    ctx.close()


def test_eight_threads_one_context_each_over_shared_segments(table):
    n, cols, exp = table
    owner = native.Context(0)
    seg = native.DeviceSegment(owner, [c.native() for c in cols])
    ctxs = [native.Context(0) for _ in range(N_THREADS)]
    _run_threads(lambda t: (ctxs[t], seg, cols, n, exp, 3))
    for c in ctxs:
        c.close()
    seg.close()
    owner.close()


def test_graph_capture_is_exclusive_on_a_shared_context(table):
    n, cols, exp = table
    ctx = native.Context(0)
    seg = native.DeviceSegment(ctx, [c.native() for c in cols])
    errors = []
    barrier = threading.Barrier(N_THREADS)
    ths = [threading.Thread(target=_worker, args=(t, ctx, seg, cols, n, exp, 2, errors, barrier, ctx)) for t in range(N_THREADS)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in ths), "a worker thread hangs"
    assert not errors, errors
    seg.close()
    ctx.close()


def test_capture_end_belongs_to_the_capturing_thread():
    L = native.load()
    import ctypes as C
    ctx = native.Context(0)
    assert L.imm3_ctx_capture_begin(ctx._h) == 0
    rc = []
    t = threading.Thread(target=lambda: rc.append(L.imm3_ctx_capture_end(ctx._h, C.byref(C.c_void_p()))))
    t.start()
    t.join(timeout=60)
    assert rc == [native.ERR_STATE]
    g = C.c_void_p()
    assert L.imm3_ctx_capture_end(ctx._h, C.byref(g)) == 0
    assert L.imm3_graph_destroy(g) == 0
    ctx.close()


def test_graph_goes_stale_when_a_recorded_query_moves_its_buffers(table):
    """a recorded graph keeps the output pointers of capture time; growing the reservation afterwards
    must not leave the graph writing through them."""
    n, cols, exp = table
    ctx = native.Context(0)
    seg = native.DeviceSegment(ctx, [c.native() for c in cols])
    name, used, sels, proj, limit = _shapes(n)[1]
    q = native.DeviceQuery(ctx, seg, used, sels, proj, limit, 1024)
    q.reserve_rows(16)                      # far too small
    q.run()
    with ctx.capture() as cap:
        q.run()
    cap.graph.launch()
    idx, vals = q.fetch_rows()              # grows the buffers and emits again
    assert idx.tolist() == exp[name]["rows"].tolist()
    with pytest.raises(native.Imm3Error) as ei:
        cap.graph.launch()
    assert ei.value.code == native.ERR_STATE
    cap.graph.close()
    q.close()
    seg.close()
    ctx.close()
