"""GPU suite: `limit` stops the scan.  ProjectIterator.hasNext returns false once `limit` rows have been emitted
(engine/src/main/scala/immutabledb/engine/operator/Project.scala:73-80) and the reference's per-segment workers then stall on the
bounded result queue (engine/Engine.scala:166,253-258): batches behind the limit are never scanned.  Here a projection with a limit
over one uniform segment runs its select launch as chunks of growing size (they end at tiles 1024, 8192, 32 768, ...); every chunk first looks at the
rows selected so far -- a device word -- and leaves at once when the limit has been reached; the offsets scan and the gather stop at the
scanned prefix.  Checked: the rows are the first `limit` survivors in row order (numpy, and the C oracle's ProjectOp), for limits that
are met in the first chunk, across a chunk boundary, in the last chunk and never; the segment's count and bitmap stay exact when a
caller asks for them (the whole select runs then); a recorded graph replays the chunks; the chunks after the limit do not read."""
import numpy as np
import pytest

from conftest import DENSE_INT, DENSE_STRING, DENSE_TINYINT, GT, LT, MATCH, RawColumn, blocks_of
from immutable3_amd import native, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def seg13(ctx):
    n = 13_100 * 1024 - 333                     # 13 100 tiles: chunks of 1024, 7168 and 4908 tiles
    ids = np.arange(n, dtype=np.int32)
    age = synth.uniform_below(33, n, 100, np.int8)
    st = synth.state_codes(35, n)
    br = blocks_of(n, 1024)
    cols = [RawColumn(DENSE_INT, 4, ids, br), RawColumn(DENSE_TINYINT, 1, age, br), RawColumn(DENSE_STRING, 2, st, br)]
    seg = native.DeviceSegment(ctx, [c.native() for c in cols])
    yield n, ids, age, st, cols, seg
    seg.close()


CASES = {
    # name: used, sels, proj, keep(ids, age, st)
    "every other tenth row": ([1, 0], [(0, GT, 89.0)], [1, 0], lambda i, a, s: a > 89),
    "second half of the sorted key": ([0, 1], [(0, GT, 7_000_000.0)], [0, 1], lambda i, a, s: i > 7_000_000),
    "last chunk only": ([0], [(0, GT, 13_000_000.0)], [0], lambda i, a, s: i > 13_000_000),
    "no survivor": ([1, 0], [(0, GT, 100.0)], [1], lambda i, a, s: np.zeros(i.shape[0], bool)),
    "string match, three columns": ([2, 0, 1], [(0, MATCH, [b"CA"])], [1, 0, 2], lambda i, a, s: (s[:, 0] == ord("C")) & (s[:, 1] == ord("A"))),
    "half of the rows": ([1, 0], [(0, GT, 49.0), (1, GT, 5.0)], [1, 0], lambda i, a, s: (a > 49) & (i > 5)),
    "one row in five thousand": ([1, 2, 0], [(0, GT, 98.0), (1, MATCH, [b"CA"])], [2, 0, 1], lambda i, a, s: (a > 98) & (s[:, 0] == ord("C")) & (s[:, 1] == ord("A"))),
}


@pytest.mark.parametrize("name", list(CASES))
def test_rows_are_the_first_survivors_and_count_and_bitmap_stay_exact(ctx, seg13, name):
    n, ids, age, st, cols, seg = seg13
    data = [ids, age, st]
    used, sels, proj, keepf = CASES[name]
    keep = keepf(ids, age, st)
    rows = np.flatnonzero(keep)
    for limit in (1, 10, 1000, 4096, 4097, 1_000_000, 5_000_000):     # (up to 4096 rows: the offsets scan and the gather are one launch, k_limit_gather)
        q = native.DeviceQuery(ctx, seg, used, sels, proj, limit)
        want = rows[:limit]
        for rnd in range(2):
            q.run()
            idx, vals = q.fetch_rows()
            assert idx.size == want.size and (idx == want).all(), (name, limit, rnd)
            for j, pj in enumerate(proj):
                assert vals[j].tobytes() == np.ascontiguousarray(data[used[pj]][want]).tobytes(), (name, limit, rnd, j)
        assert q.row_count() == want.size
        assert q.count() == rows.size, (name, limit)                  # the whole select runs now: the segment's count, not the prefix's
        assert q.bitmap().tobytes() == np.packbits(keep, bitorder="little").tobytes()[: q.total_words * 8].ljust(q.total_words * 8, b"\0"), (name, limit)
        idx, _ = q.fetch_rows()
        assert (idx == want).all(), (name, limit)
        q.run()                                                          # and a chunked run again behind the whole one
        idx, _ = q.fetch_rows()
        assert (idx == want).all(), (name, limit)
        q.close()


def test_against_the_oracle_and_in_a_graph(ctx, oracle):
    rng = np.random.default_rng(11)
    n = 2_300_000 + 77                                                  # 2247 tiles: chunks of 1024 and 1223
    ids = rng.integers(0, 1000, size=n).astype(np.int32)
    age = rng.integers(0, 100, size=n).astype(np.int8)
    br = blocks_of(n, 1024)
    cols = [RawColumn(DENSE_INT, 4, ids, br), RawColumn(DENSE_TINYINT, 1, age, br)]
    seg = native.DeviceSegment(ctx, [c.native() for c in cols])
    sels = [(0, GT, 18.0), (0, LT, 30.0), (1, GT, 100.0)]
    ocols = [cols[1].ocol(), cols[0].ocol()]
    words, total = oracle.scan_select(ocols, sels, 1024)
    for limit in (10, 1500, 4096, 200_000):
        cnt, batch, pos, ovals, _ = oracle.project(ocols, [1, 0], limit, 1024, words)
        pos = batch.astype(np.int64) * 1024 + pos                     # (batch, position in the batch) -> row of the segment: every block holds 1024 rows
        q = native.DeviceQuery(ctx, seg, [1, 0], sels, [1, 0], limit)
        q.run()
        idx, vals = q.fetch_rows()
        assert idx.size == cnt and (idx == pos).all(), limit
        assert all(a.tobytes() == b.tobytes() for a, b in zip(vals, ovals)), limit
        with ctx.capture() as cap:
            q.run()
        for _ in range(3):
            cap.graph.launch()
        idx, vals = q.fetch_rows()
        assert idx.size == cnt and (idx == pos).all() and all(a.tobytes() == b.tobytes() for a, b in zip(vals, ovals)), limit
        assert q.count() == total
        cap.graph.close()
        q.close()
    seg.close()


def test_chunks_behind_the_limit_do_not_touch_the_bitmap(ctx, seg13):
    """`select id ... where id > 5 limit 10`: met in the first chunk (1024 tiles); the chunks behind it leave at once.  Shown on the
    device words themselves: the scanned-tile word stops at the first chunk, and bitmap lines behind it -- poisoned before the run --
    are still poisoned after it.  With the survivors in the second half of the sorted key the scan goes on until the chunk that
    holds them."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    n, ids, age, st, cols, seg = seg13
    n_tiles = (n + 1023) // 1024
    for thr, want_scanned in ((5.0, 1024), (3_000_000.0, 8192), (9_000_000.0, n_tiles)):   # chunks end at 1024, 8192, 13 100: row 3 M = tile 2929 sits in the second, row 9 M = tile 8789 in the last
        q = native.DeviceQuery(ctx, seg, [0], [(0, GT, thr)], [0], 10)
        q.run()                                                              # (allocates everything)
        ctx.sync()
        poison = np.full(n_tiles * 16, 0xA5A5A5A5A5A5A5A5, np.uint64)
        assert hip.hipMemcpy(C.c_void_p(q.device_ptr(0)), C.c_void_p(poison.ctypes.data), C.c_size_t(poison.nbytes), C.c_int(1)) == 0
        q.run()
        ctx.sync()
        head = np.zeros(13, np.uint64)
        assert hip.hipMemcpy(C.c_void_p(head.ctypes.data), C.c_void_p(q.device_ptr(1)), C.c_size_t(head.nbytes), C.c_int(2)) == 0
        raw = np.zeros(n_tiles * 16, np.uint64)
        assert hip.hipMemcpy(C.c_void_p(raw.ctypes.data), C.c_void_p(q.device_ptr(0)), C.c_size_t(raw.nbytes), C.c_int(2)) == 0
        scanned = int(head[12])
        assert scanned == want_scanned, (thr, scanned, want_scanned)
        assert (raw[scanned * 16:] == 0xA5A5A5A5A5A5A5A5).all(), thr        # untouched behind the scanned prefix
        keep = ids > thr
        exp = np.packbits(keep[: scanned * 1024], bitorder="little").view(np.uint8)
        assert raw[: scanned * 16].view(np.uint8)[: exp.size].tobytes() == exp.tobytes(), thr
        idx, _ = q.fetch_rows()
        assert (idx == np.flatnonzero(keep)[:10]).all(), thr
        q.close()


def test_device_side_count_consumers_get_the_segments_count(ctx, seg13):
    """The count log and the count all-reduce read the count on the device: behind a run that stopped at its limit they still get the
    segment's count (a logged query scans whole; the collective runs the whole select first), `Engine.scala:190-196`."""
    import torch
    n, ids, age, st, cols, seg = seg13
    want = int((age > 89).sum())
    q = native.DeviceQuery(ctx, seg, [1, 0], [(0, GT, 89.0)], [1, 0], 10)
    q.run()
    assert q.row_count() == 10
    comm = native.Comm(ctx, 1, 0, native.comm_unique_id())
    try:
        assert comm.allreduce_count([q]) == want
        q.run()                                                  # (stops early again)
        assert comm.allreduce_count([q]) == want
    finally:
        comm.close()
    log = torch.zeros(4, dtype=torch.int64, device="cuda:0")
    q.log_counts(log.data_ptr(), 4)
    for _ in range(3):
        q.run()
    ctx.sync()
    assert log.cpu().tolist() == [want, want, want, 0]
    q.log_counts(0, 0)
    idx, _ = q.fetch_rows()
    assert (idx == np.flatnonzero(age > 89)[:10]).all()
    q.close()
