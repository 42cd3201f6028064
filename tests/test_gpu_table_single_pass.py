"""GPU suite: the one-launch projection over a TABLE (csrc/imm3_project_table.hip: the TABLE instances of k_filter_project) --
ScanOp -> SelectOp* -> ProjectOp of every segment a GPU owns in ONE launch over the tile table.  The reference fans out one
pipeline per segment and merges their rows into one result (engine/src/main/scala/immutabledb/engine/Engine.scala:176-196;
its on-disk shape is ~98 loader-made segments per 100 M rows, README.md:10); ProjectIterator.next's walk per batch is
Project.scala:37-64.  Through the C ABI, bit-exact: per-segment bitmaps, the count, rows in (segment, row) order, values.

Covered: survivors that straddle segment boundaries and every segment's partial last tile (loader-quirk segments of S * B + 1
rows, a short segment, a one-row and an empty one) at sparse / dense / full fills and for a clustered range of the sorted key
(ranges that outgrow their LDS ring are unpacked from the source columns through the tile descriptors; fully surviving ranges
are copied tile by tile, across a segment boundary with two base pointers), several rounds of spans, every column-kind shape the
planner gives a table, a reservation that is too small, graph replays, the oracle at a smaller size, and the plan the library
makes by itself (cost model on the table's sample)."""
import numpy as np
import pytest

from conftest import DENSE_INT, DENSE_STRING, DENSE_TINYINT, GT, LT, MATCH, RawColumn, blocks_of
from immutable3_amd import native, synth

pytestmark = pytest.mark.gpu
CODES = [b"CA", b"NY", b"TX", b"WA", b"VA", b"DC", b"CT"]


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0)
    yield c
    c.close()


def loader_blocks(n, block=1024):
    """Block sizes of a segment of n rows: full blocks, then the rest (n = S * B + 1: the loader's trailing one-row block)."""
    return blocks_of(n, block)


class Table:
    """A table of segments with columns id:int32 (sorted over the whole table), key:int32 (random), age:int8, tiny:int8, st:string(2)."""

    def __init__(self, ctx, seg_rows, seed=7):
        self.ctx = ctx
        self.seg_rows = list(seg_rows)
        rng = np.random.default_rng(seed)
        self.cols, self.dsegs = [], []
        first = 0
        for n in self.seg_rows:
            ids = (np.arange(n, dtype=np.int64) + first).astype(np.int32)
            key = rng.integers(0, 2 ** 30, size=n).astype(np.int32)
            age = rng.integers(0, 100, size=n).astype(np.int8)
            tiny = rng.integers(-128, 128, size=n).astype(np.int8)
            st = np.array([list(c) for c in CODES], dtype=np.uint8)[rng.integers(0, len(CODES), size=n)].reshape(n, 2)
            br = loader_blocks(n)
            cols = [RawColumn(DENSE_INT, 4, ids, br), RawColumn(DENSE_INT, 4, key, br), RawColumn(DENSE_TINYINT, 1, age, br),
                    RawColumn(DENSE_TINYINT, 1, tiny, br), RawColumn(DENSE_STRING, 2, st, br)]
            self.cols.append(cols)
            self.dsegs.append(native.DeviceSegment(ctx, [c.native() for c in cols]))
            first += n
        self.table = native.DeviceTable(ctx, self.dsegs)
        self.data = [[c.values for c in cols] for cols in self.cols]

    def column(self, u):
        parts = [d[u] for d in self.data]
        return np.concatenate(parts) if parts else np.zeros(0)

    def close(self):
        self.table.close()
        for d in self.dsegs:
            d.close()


def keep_of(t, used, sels):
    """numpy evaluation of the conjunction over the whole table (rows of all segments concatenated)."""
    n = sum(t.seg_rows)
    keep = np.ones(n, bool)
    for i, op, v in sels:
        col = t.column(used[i])
        if op == GT:
            keep &= col > v
        elif op == LT:
            keep &= col < v
        else:
            codes = col.view("<u2").reshape(-1)
            want = np.array([int.from_bytes(x, "little") for x in v], dtype=np.uint16)
            keep &= np.isin(codes, want)
    return keep


def check_table_query(t, q, used, sels, proj, tag):
    keep = keep_of(t, used, sels)
    starts = np.concatenate([[0], np.cumsum(t.seg_rows)]).astype(np.int64)
    assert q.count() == int(keep.sum()), tag
    words = q.bitmap()
    fb, fw = q.segment_starts()
    for si, n in enumerate(t.seg_rows):
        k = keep[starts[si]: starts[si + 1]]
        want = np.packbits(k, bitorder="little")
        want = np.concatenate([want, np.zeros((-want.size) % 8, np.uint8)]).view("<u8")
        w0, w1 = int(fw[si]), int(fw[si + 1])
        assert words[w0: w0 + want.size].tolist() == want.tolist(), (tag, "bitmap of segment", si)
        assert not words[w0 + want.size: w1].any(), (tag, "padding behind segment", si)
    idx, vals = q.fetch_rows()
    rows = np.flatnonzero(keep)
    assert idx.size == rows.size, (tag, idx.size, rows.size)
    seg_of, row_of = q.locate_rows(idx)
    want_seg = np.searchsorted(starts, rows, side="right") - 1
    assert (seg_of == want_seg).all() and (row_of == rows - starts[want_seg]).all(), tag
    for j, pj in enumerate(proj):
        col = t.column(used[pj])
        assert vals[j].tobytes() == np.ascontiguousarray(col[rows]).tobytes(), (tag, "column", j)


def pinned(variant, ctx, *args):
    """A query created under a tuning variant (12: the plan made at creation stands; 200 + P: the one launch at that P)."""
    ctx.set_tuning(variant, 0)
    try:
        return native.DeviceQuery(ctx, *args)
    finally:
        ctx.set_tuning(0, 0)


# loader-quirk segments (S * B + 1 rows: a full block layout and a trailing one-row block), one short one,
# a one-row and an empty segment: eight partial tiles in the middle of the table
SEG_ROWS = [1000 * 1024 + 1] * 6 + [123_457, 1, 0, 700 * 1024 + 1]


@pytest.fixture(scope="module")
def big(ctx):
    t = Table(ctx, SEG_ROWS)
    yield t
    t.close()


FILLS = {"sparse": (96.5, 0.97 * 2 ** 30), "some": (88.0, 0.5 * 2 ** 30), "dense": (35.0, 0.1 * 2 ** 30), "full": (-1.0, -1.0)}


@pytest.mark.parametrize("fill", list(FILLS))
@pytest.mark.parametrize("P", [2, 5])
def test_table_one_launch_rows_straddle_segments_and_partial_tiles(ctx, big, fill, P):
    """select key, age where age > a and key > k over ten segments: ONE launch; rows, bitmap and count against numpy.  P = 2: 16
    tiles per span -> ~470 spans, two rounds for one work-group per CU, ranges shorter than a segment's tail; P = 5: ranges that
    hold a partial tile AND the next segment's first tiles."""
    a, k = FILLS[fill]
    used, sels, proj = [2, 1], [(0, GT, a), (1, GT, k)], [1, 0]
    q = pinned(200 + P, ctx, big.table, used, sels, proj, 0, 1024)
    p = q.plan()
    assert p["single_pass"] and p["P"] == P, p
    q.run()
    check_table_query(big, q, used, sels, proj, (fill, P, "first run"))
    assert q.plan()["ran_single_pass"] and q.plan()["abandoned_runs"] == 0 and q.plan()["busy_runs"] == 0, q.plan()
    q.run()
    check_table_query(big, q, used, sels, proj, (fill, P, "second run"))
    assert q.plan()["ran_single_pass"]
    q.close()


@pytest.mark.parametrize("lo_seg,hi_seg", [(2, 4), (0, 9), (5, 6)])
def test_table_one_launch_clustered_range_of_the_sorted_key(ctx, big, lo_seg, hi_seg):
    """id in (middle of segment lo, middle of segment hi): every row of whole segments survives -- fully surviving ranges are copied
    tile by tile with each tile's own pointer (a range straddles two segments), the partial tiles take the general walk."""
    starts = np.concatenate([[0], np.cumsum(big.seg_rows)])
    lo = float(starts[lo_seg] + big.seg_rows[lo_seg] // 2)
    hi = float(starts[hi_seg] + max(big.seg_rows[hi_seg] // 2, 1))
    for used, sels, proj in (([0], [(0, GT, lo), (0, LT, hi)], [0]), ([0, 2], [(0, GT, lo), (0, LT, hi), (1, GT, -1.0)], [0, 1]),
                             ([0, 4], [(0, GT, lo), (0, LT, hi), (1, MATCH, CODES)], [1, 0])):
        q = pinned(12, ctx, big.table, used, sels, proj, 0, 1024)
        assert q.plan()["single_pass"], q.plan()
        q.run()
        check_table_query(big, q, used, sels, proj, ("clustered", lo_seg, hi_seg, used))
        q.run()                                        # (the host has seen the count: P may have changed)
        check_table_query(big, q, used, sels, proj, ("clustered again", lo_seg, hi_seg, used))
        assert q.plan()["ran_single_pass"], q.plan()
        q.close()


# predicate columns (indices into [id, key, age, tiny, st]) of every column-kind instance
SHAPES = [[1], [2], [4], [0, 1], [1, 2], [2, 3], [1, 4], [2, 4], [0, 1, 2], [1, 2, 3], [2, 3, 4], [0, 1, 4], [1, 2, 4]]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "-".join(map(str, s)))
def test_table_one_launch_every_kind(ctx, big, shape):
    thr = {0: float(sum(big.seg_rows) // 3), 1: 0.6 * 2 ** 30, 2: 70.0, 3: 40.0}
    used = list(shape)
    sels = [(i, MATCH, CODES[:3]) if u == 4 else (i, GT, thr[u]) for i, u in enumerate(used)]
    proj = list(range(len(used)))[::-1]
    q = pinned(12, ctx, big.table, used, sels, proj, 0, 1024)
    assert q.plan()["single_pass"], q.plan()
    q.run()
    check_table_query(big, q, used, sels, proj, ("kinds", shape))
    assert q.plan()["ran_single_pass"], q.plan()
    q.close()


def test_table_one_launch_reservation_too_small_and_graph_replay(ctx, big):
    used, sels, proj = [2, 1], [(0, GT, 80.0), (1, GT, 0.2 * 2 ** 30)], [1, 0]
    q = pinned(203, ctx, big.table, used, sels, proj, 0, 1024)
    q.reserve_rows(1000)                               # far too few: the rows are gathered again from the bitmap when fetched
    q.run()
    check_table_query(big, q, used, sels, proj, "small reservation")
    q.close()
    q = pinned(203, ctx, big.table, used, sels, proj, 0, 1024)
    q.reserve_rows(sum(big.seg_rows))
    q.run()
    ctx.sync()
    with ctx.capture() as cap:
        q.run()
    for i in range(3):
        cap.graph.launch()
        check_table_query(big, q, used, sels, proj, ("replay", i))
        assert q.plan()["ran_single_pass"], q.plan()
    cap.graph.close()
    q.close()


def test_table_planner_takes_the_one_launch_by_itself(ctx, big):
    """No tuning: C3's shape over the table (7.3 M rows, ~10 % survivors).  Whatever the cost model picks, the rows are right; the
    plan it reports must be one the table path has (the one launch, or the bitmap path: no records)."""
    used, sels, proj = [2, 0], [(0, GT, 18.0), (0, LT, 30.0), (1, GT, 1000.0)], [1, 0]
    q = native.DeviceQuery(ctx, big.table, used, sels, proj, 0, 1024)
    assert not q.plan()["records"], q.plan()
    q.run()
    check_table_query(big, q, used, sels, proj, "planner")
    q.run()
    check_table_query(big, q, used, sels, proj, "planner, second run")
    q.close()
    # a gathered SELECT-list column keeps the bitmap path
    q = native.DeviceQuery(ctx, big.table, [2, 0, 4], [(0, GT, 90.0)], [1, 2, 0], 0, 1024)
    assert not q.plan()["single_pass"] and not q.plan()["records"], q.plan()
    q.run()
    check_table_query(big, q, [2, 0, 4], [(0, GT, 90.0)], [1, 2, 0], "gathered")
    q.close()


SMALL = [
    [8 * 1024 + 1, 8 * 1024 + 1, 3 * 1024 + 700],
    [100, 0, 1, 64, 5000],
    [70000, 1024, 2048, 1025],
    [40 * 1024 + 1] * 5,
]


@pytest.mark.parametrize("seg_rows", SMALL, ids=lambda s: "x".join(map(str, s)))
def test_table_one_launch_against_the_oracle(ctx, oracle, seg_rows):
    """Small tables, the one launch pinned, against the C oracle per segment: bitmap words, count, layout, rows in (segment, row)
    order, values."""
    t = Table(ctx, seg_rows, seed=len(seg_rows) + seg_rows[0])
    try:
        for used, sels, proj in (([2, 1], [(0, GT, 60.0), (1, GT, 0.3 * 2 ** 30)], [1, 0]), ([4, 3], [(0, MATCH, CODES[:2]), (1, LT, 0.0)], [0, 1]),
                                 ([0], [], [0]), ([3], [(0, GT, -128.0)], [0])):
            q = pinned(12, ctx, t.table, used, sels, proj, 0, 1024)
            one_launch = len(sels) > 0                 # (no predicate column: the SELECT list is gathered -- the bitmap path)
            assert q.plan()["single_pass"] == one_launch, (used, sels, q.plan())
            q.run()
            words, count = q.bitmap(), q.count()
            size, oid, woff = q.batches()
            fb, fw = q.segment_starts()
            idx, vals = q.fetch_rows()
            seg_of, row_of = q.locate_rows(idx)
            assert q.plan()["ran_single_pass"] == one_launch, q.plan()
            q.close()
            exp, total = [], 0
            for si, cols in enumerate(t.cols):
                ucols = [cols[i] for i in used]
                ow, oc = oracle.scan_select([c.ocol() for c in ucols], sels, 1024, 1)
                osize, ooid, owoff, _ = oracle.layout(ucols[0].ocol(), 1024)
                total += oc
                b0, b1 = int(fb[si]), int(fb[si + 1])
                assert size[b0:b1].tolist() == osize.tolist() and oid[b0:b1].tolist() == ooid.tolist()
                assert words[int(fw[si]): int(fw[si]) + ow.size].tolist() == ow.tolist()
                assert not words[int(fw[si]) + ow.size: int(fw[si + 1])].any()
                n, batch, pos, ovals, _ = oracle.project([c.ocol() for c in ucols], proj, 0, 1024, ow)
                starts = np.concatenate([[0], np.cumsum(osize.astype(np.int64))])
                for r in range(n):
                    exp.append((si, int(starts[batch[r]] + pos[r]), [bytes(v[r]) for v in ovals]))
            assert count == total
            assert idx.shape[0] == len(exp)
            assert seg_of.tolist() == [e[0] for e in exp] and row_of.tolist() == [e[1] for e in exp]
            for j in range(len(proj)):
                assert [bytes(v) for v in vals[j]] == [e[2][j] for e in exp]
    finally:
        t.close()
