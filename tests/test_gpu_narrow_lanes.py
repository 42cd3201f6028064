"""GPU suite: the IN-LANE instances of k_filter_tile (csrc/imm3_kernels.hip: lane_tile()) -- select chains over int8 and 2-byte
string columns only (SelectIteratorRange / SelectIteratorMatch, Select.scala:25-127, over DenseCodec's raw bytes,
DenseCodec.scala:51-73), which evaluate a tile without the LDS transpose: lane l keeps 16 consecutive rows (int8 only) or two runs
of 8 rows (with a string column), builds their mask with compare + add-with-carry and four / eight neighbouring lanes assemble the
bitmap word with DPP moves.  Bit-exact against numpy and the oracle: every bit position of a lane's mask, every lane of a word's
quad / octet, both runs, every kind combination, ragged ends, the multi-pass AND (`and_existing`), parked (deferred) and direct
bitmap lines, count-only runs, tables, limit chunks."""
import numpy as np
import pytest

from conftest import DENSE_STRING, DENSE_TINYINT, GT, LT, MATCH, RawColumn, blocks_of

pytestmark = pytest.mark.gpu
CODES = [b"CA", b"NY", b"TX", b"WA", b"VA", b"DC", b"CT"]


@pytest.fixture(scope="module")
def ctx():
    from immutable3_amd import native
    c = native.Context(0)
    yield c
    c.close()


def words_of(keep):
    bits = np.packbits(keep, bitorder="little")
    return np.concatenate([bits, np.zeros((-bits.size) % 8, np.uint8)]).view("<u8")


def narrow_columns(rng, n):
    a = [rng.integers(-128, 128, size=n).astype(np.int8) for _ in range(4)]
    st = np.array([list(c) for c in CODES], np.uint8)[rng.integers(0, len(CODES), size=n)]
    return a, st


def raw_cols(a, st, br):
    return [RawColumn(DENSE_TINYINT, 1, x, br) for x in a] + [RawColumn(DENSE_STRING, 2, st, br)]


def is_code(st, code):
    return (st[:, 0] == code[0]) & (st[:, 1] == code[1])


# (used columns of [a0, a1, a2, a3, st], select list, numpy mask)
CASES = {
    "I8": ([0], lambda s: [(0, GT, -100.0), (0, LT, 90.0)], lambda a, st: (a[0] > -100) & (a[0] < 90)),
    "I8 one-sided": ([1], lambda s: [(0, GT, 17.0)], lambda a, st: a[1] > 17),
    "I8+I8": ([0, 1], lambda s: [(0, GT, -64.0), (1, LT, 100.0)], lambda a, st: (a[0] > -64) & (a[1] < 100)),
    "I8+I8+I8": ([0, 1, 2], lambda s: [(0, GT, -64.0), (1, LT, 100.0), (2, GT, -127.0), (2, LT, 127.0)],
                 lambda a, st: (a[0] > -64) & (a[1] < 100) & (a[2] > -127) & (a[2] < 127)),
    "S2": ([4], lambda s: [(0, MATCH, [b"CA"])], lambda a, st: is_code(st, b"CA")),
    "S2 in(3)": ([4], lambda s: [(0, MATCH, [b"CA", b"DC", b"WA"])], lambda a, st: is_code(st, b"CA") | is_code(st, b"DC") | is_code(st, b"WA")),
    "I8+S2": ([0, 4], lambda s: [(0, GT, -90.0), (1, MATCH, [b"NY", b"TX"])], lambda a, st: (a[0] > -90) & (is_code(st, b"NY") | is_code(st, b"TX"))),
    "I8+I8+S2": ([0, 3, 4], lambda s: [(0, GT, -90.0), (1, LT, 111.0), (2, MATCH, [b"VA"])], lambda a, st: (a[0] > -90) & (a[3] < 111) & is_code(st, b"VA")),
    # four and five narrow columns: two passes -- the second ANDs into the first's words (and_existing), in both lane layouts
    "I8 x4": ([0, 1, 2, 3], lambda s: [(0, GT, -100.0), (1, LT, 100.0), (2, GT, -110.0), (3, LT, 120.0)],
              lambda a, st: (a[0] > -100) & (a[1] < 100) & (a[2] > -110) & (a[3] < 120)),
    "I8 x4 + S2": ([0, 1, 2, 3, 4], lambda s: [(0, GT, -100.0), (1, LT, 100.0), (2, GT, -110.0), (3, LT, 120.0), (4, MATCH, [b"CA", b"NY", b"TX", b"WA"])],
                   lambda a, st: (a[0] > -100) & (a[1] < 100) & (a[2] > -110) & (a[3] < 120) & (is_code(st, b"CA") | is_code(st, b"NY") | is_code(st, b"TX") | is_code(st, b"WA"))),
}


@pytest.mark.parametrize("n", [1, 63, 1024, 1025, 16 * 1024 + 777, 700 * 1024 + 5, 3_000_000 + 11])
def test_every_narrow_combination_against_numpy(ctx, n):
    """Sizes: below a tile (the rolled partial-tile path only), whole tiles + a ragged end, a few hundred tiles (lines stored
    directly) and thousands (lines parked in LDS, stored in bursts)."""
    from immutable3_amd import native
    rng = np.random.default_rng(n)
    a, st = narrow_columns(rng, n)
    cols = raw_cols(a, st, blocks_of(n, 1024))
    seg = native.DeviceSegment(ctx, [c.native() for c in cols])
    for name, (used, sels, mask) in CASES.items():
        keep = mask(a, st)
        q = native.DeviceQuery(ctx, seg, used, sels(None))
        q.run()
        assert q.count() == int(keep.sum()), (name, n)
        got = q.bitmap()
        want = words_of(keep)
        assert got[: want.size].tolist() == want.tolist(), (name, n)
        assert not got[want.size:].any(), (name, n)
        q.run_count()                                                   # the instance that stores no line
        assert q.count() == int(keep.sum()), (name, n, "count-only")
        q.close()
    seg.close()


def test_single_bits_land_where_they_belong(ctx):
    """One surviving row at a time, at every position of a tile's first 128 rows and of rows 448..639 (both runs of the split layout,
    every lane of a quad / octet, every bit of a lane's mask): the bit assembly cannot hide behind dense random data."""
    from immutable3_amd import native
    n = 3 * 1024
    positions = list(range(128)) + list(range(448, 640)) + [1023, 1024, 2047, 3071]
    for pos in positions:
        a0 = np.full(n, -5, np.int8)
        a0[pos] = 77
        st = np.tile(np.frombuffer(b"NY", np.uint8), (n, 1)).copy()
        st[pos] = np.frombuffer(b"CA", np.uint8)
        cols = [RawColumn(DENSE_TINYINT, 1, a0, blocks_of(n, 1024)), RawColumn(DENSE_STRING, 2, st, blocks_of(n, 1024))]
        seg = native.DeviceSegment(ctx, [c.native() for c in cols])
        for used, sels in (([0], [(0, GT, 0.0)]), ([1], [(0, MATCH, [b"CA"])]), ([0, 1], [(0, GT, 0.0), (1, MATCH, [b"CA", b"TX"])])):
            q = native.DeviceQuery(ctx, seg, used, sels)
            q.run()
            w = q.bitmap()
            assert q.count() == 1 and int(w[pos // 64]) == 1 << (pos % 64) and np.count_nonzero(w) == 1, (pos, used)
            q.close()
        seg.close()


def test_against_the_oracle_with_a_projection(ctx, oracle):
    """The same kinds through the whole path -- select chain, offsets scan, gather -- against the oracle's scan_select + project
    (ragged blocks: the loader's trailing one-row block)."""
    from test_gpu_parity import check
    rng = np.random.default_rng(77)
    n = 40 * 1024 + 1
    a, st = narrow_columns(rng, n)
    cols = raw_cols(a, st, blocks_of(n, 1024))
    check(ctx, oracle, cols, [0, 4], [(0, GT, 100.0), (1, MATCH, [b"CA"])], proj=[1, 0])
    check(ctx, oracle, cols, [1], [(0, GT, 120.0)], proj=[0])
    check(ctx, oracle, cols, [4], [(0, MATCH, [b"DC", b"CT"])], proj=[0], limit=50)
    check(ctx, oracle, cols, [0, 1, 2], [(0, GT, 0.0), (1, GT, 0.0), (2, GT, 0.0)], proj=[2, 0])


def test_tables_and_limit_chunks(ctx):
    """Table queries (one partial tile per segment, lines parked with their tile numbers) and a limit scan's chunks over the in-lane
    instances."""
    from immutable3_amd import native
    rng = np.random.default_rng(5)
    rows = [70_001, 1024, 333, 250_000]
    parts = [narrow_columns(rng, m) for m in rows]
    segs = [native.DeviceSegment(ctx, [c.native() for c in raw_cols(a, st, blocks_of(m, 1024))]) for (a, st), m in zip(parts, rows)]
    table = native.DeviceTable(ctx, segs)
    for name in ("I8", "I8+I8", "S2", "I8+S2", "I8+I8+S2", "I8 x4", "I8 x4 + S2"):  # (the last two: a second pass ANDs into the first's words)
        used, sels, mask = CASES[name]
        q = native.DeviceQuery(ctx, table, used, sels(None), (), 0, 1024)
        q.run()
        want = sum(int(mask(a, st).sum()) for a, st in parts)
        assert q.count() == want, name
        words = q.bitmap()
        _, fw = q.segment_starts()
        for si, ((a, st), m) in enumerate(zip(parts, rows)):  # a table's bitmap: every segment's words from its own start, zero padding behind
            want_w = words_of(mask(a, st))
            w0, w1 = int(fw[si]), int(fw[si + 1])
            assert words[w0: w0 + want_w.size].tolist() == want_w.tolist(), (name, si)
            assert not words[w0 + want_w.size: w1].any(), (name, si)
        q.close()
    table.close()
    for s in segs:
        s.close()
    # a limit scan: chunks of growing size, each an in-lane launch that adds to the running count
    n = 9_000_000
    a0 = rng.integers(-128, 128, size=n).astype(np.int8)
    seg = native.DeviceSegment(ctx, [RawColumn(DENSE_TINYINT, 1, a0, blocks_of(n, 1024)).native()])
    rows_ = np.flatnonzero(a0 > 125)
    for limit in (10, 5000):
        q = native.DeviceQuery(ctx, seg, [0], [(0, GT, 125.0)], [0], limit)
        q.run()
        idx, vals = q.fetch_rows()
        assert idx.tolist() == rows_[:limit].tolist() and vals[0].view(np.int8).reshape(-1).tolist() == a0[rows_[:limit]].tolist()
        assert q.count() == rows_.size
        q.close()
    seg.close()
