"""Parity tests proper: the HIP path, called through the C ABI (immutable3_amd.native -> libimm3.so),
against the CPU oracle on the same seeded inputs.  Bit-exact: selection bitmap words, selected-row
count, emitted row order and projected values."""
import numpy as np
import pytest

from conftest import DENSE_INT, DENSE_STRING, DENSE_TINYINT, EQ, GT, LT, MATCH, NOOP, NOTMATCH, RawColumn, blocks_of

pytestmark = pytest.mark.gpu

CODES = [b"CA", b"NY", b"TX", b"WA", b"VA", b"DC", b"CT"]


@pytest.fixture(scope="module")
def ctx():
    from immutable3_amd import native
    assert native.device_count() >= 1, "no HIP device: the GPU path has no CPU fallback"
    c = native.Context(0)
    yield c
    c.close()


def gpu_run(ctx, cols, used, sels, proj=(), limit=0, block_size=1024, reserve=None):
    from immutable3_amd import native
    seg = native.DeviceSegment(ctx, [c.native() for c in cols])
    q = native.DeviceQuery(ctx, seg, used, sels, proj, limit, block_size)
    if reserve is not None:
        q.reserve_rows(reserve)
    q.run()
    out = {"words": q.bitmap(), "count": q.count(), "layout": q.batches(), "n_batches": q.n_batches}
    if proj:
        idx, vals = q.fetch_rows()
        out["row_index"], out["vals"] = idx, vals
    q.close()
    seg.close()
    return out


def check(ctx, oracle, cols, used, sels, proj=(), limit=0, block_size=1024, reserve=None):
    ucols = [cols[i] for i in used]
    ow, oc = oracle.scan_select([c.ocol() for c in ucols], sels, block_size, 1)
    g = gpu_run(ctx, cols, used, sels, proj, limit, block_size, reserve)
    assert g["count"] == oc
    assert g["words"].tolist() == ow.tolist()
    size, oid, woff, _ = oracle.layout(ucols[0].ocol(), block_size)
    assert g["layout"][0].tolist() == size.tolist()
    assert g["layout"][1].tolist() == oid.tolist()
    assert g["layout"][2].tolist() == woff.tolist()
    if proj:
        n, batch, pos, vals, _ = oracle.project([c.ocol() for c in ucols], list(proj), limit, block_size, ow)
        assert g["row_index"].shape[0] == n
        starts = np.concatenate([[0], np.cumsum(size.astype(np.int64))])
        assert g["row_index"].astype(np.int64).tolist() == (starts[batch] + pos).tolist()
        for gv, ov in zip(g["vals"], vals):
            assert gv.tobytes() == ov.tobytes()
    return g


def make_cols(rng, n, block_rows, small_ids=False):
    ids = rng.integers(-2**31, 2**31, size=n, dtype=np.int64).astype(np.int32)
    if small_ids:
        ids = rng.integers(-50, 50, size=n).astype(np.int32)
    age = rng.integers(-128, 128, size=n).astype(np.int8)
    st = np.array([list(CODES[i]) for i in rng.integers(0, len(CODES), size=n)], dtype=np.uint8).reshape(n, 2)
    return [RawColumn(DENSE_INT, 4, ids, block_rows), RawColumn(DENSE_TINYINT, 1, age, block_rows),
            RawColumn(DENSE_STRING, 2, st, block_rows)]


# ---- C1: test_100 -------------------------------------------------------------------
def test_c1_test100(ctx, oracle):
    from immutable3_amd import synth
    t = synth.test_100()
    cols = [RawColumn(DENSE_INT, 4, t["id"], [100]), RawColumn(DENSE_STRING, 2, t["state"], [100]),
            RawColumn(DENSE_TINYINT, 1, t["age"], [100])]
    # select id, age from test_100 where (age > 18 and age < 30) limit 10 -> used = [age, id]
    g = check(ctx, oracle, cols, [2, 0], [(0, GT, 18.0), (0, LT, 30.0)], proj=[1, 0], limit=10)
    assert g["words"].tolist() == [0x0042100108008400, 0x0000000001080084] and g["count"] == 11
    assert g["vals"][0].view("<i4").reshape(-1).tolist() == [10, 15, 27, 32, 44, 49, 54, 66, 71, 83]
    assert g["vals"][1].view(np.int8).reshape(-1).tolist() == [21, 26, 20, 25, 19, 24, 29, 23, 28, 22]
    g = check(ctx, oracle, cols, [1], [(0, MATCH, [b"CA"])], proj=[0])
    assert g["count"] == 15 and g["row_index"].tolist() == list(range(0, 100, 7))
    g = check(ctx, oracle, cols, [2, 1, 0], [(0, GT, 18.0), (0, LT, 30.0), (1, MATCH, [b"CA"])], proj=[2, 1, 0])
    assert g["count"] == 1 and g["vals"][0].view("<i4").reshape(-1).tolist() == [49]


# ---- layouts: uniform, tails, ragged, loader quirk, empty ------------------------------------------
LAYOUTS = [
    (0, []), (1, [1]), (63, [63]), (64, [64]), (65, [65]), (100, [100]), (1024, [1024]), (1025, [1024, 1]),
    (2048 + 256, [1024, 1024, 256]), (300, [128, 128, 44]), (25, [4, 4, 1, 4, 4, 1, 4, 3]), (130, [64, 0, 66]),
    (200, [100, 100]), (5000, blocks_of(5000, 1024)), (777, blocks_of(777, 10)), (4097, [4096, 1]),
    (70000, blocks_of(70000, 1024)), (3 * 1024 + 1, [1024, 1024, 1024, 1]),
]


@pytest.mark.parametrize("n,block_rows", LAYOUTS)
def test_layouts_random_queries(ctx, oracle, n, block_rows):
    rng = np.random.default_rng(4321 + n)
    for trial in range(4):
        cols = make_cols(rng, n, block_rows, small_ids=bool(trial & 1))
        used = [int(i) for i in rng.permutation(3)[: rng.integers(1, 4)]]
        sels = []
        for _ in range(rng.integers(0, 5)):
            ci = int(rng.integers(0, len(used)))
            c = cols[used[ci]]
            if c.codec == DENSE_STRING:
                vals = [CODES[i] for i in rng.integers(0, len(CODES), size=int(rng.integers(0, 4)))]
                if rng.random() < 0.3:
                    vals.append(b"CAL")
                sels.append((ci, MATCH, vals))
            else:
                cond = [GT, LT, EQ][int(rng.integers(0, 3))]
                v = float(rng.integers(-60, 60)) if rng.random() < 0.7 else float(
                    rng.choice([200.0, 128.0, 256.0, -129.0, 3e9, -3e9, float("nan"), 2147483647.0, -2147483648.0, 17.5]))
                sels.append((ci, cond, v))
        proj = [int(i) for i in rng.permutation(len(used))[: rng.integers(0, len(used) + 1)]]
        limit = int(rng.choice([0, 0, 1, 7, 10, 10**6]))
        check(ctx, oracle, cols, used, sels, proj, limit)


# ---- C2: 1M-row DENSE_INT, RangeFilter only + threshold edge sweep -----------------------------------
def test_c2_one_million_int(ctx, oracle):
    from immutable3_amd import synth
    n = 1_000_000
    v = synth.uniform_int30(1, n)
    cols = [RawColumn(DENSE_INT, 4, v, blocks_of(n, 1024))]
    g = check(ctx, oracle, cols, [0], [(0, GT, float(2**28)), (0, LT, float(3 * 2**28))])
    assert abs(g["count"] / n - 0.5) < 0.01
    for sels in ([(0, GT, -2147483648.0)], [(0, LT, 2147483647.0)], [(0, GT, 2147483647.0)], [(0, LT, -2147483648.0)],
                 [(0, GT, float("nan"))], [(0, GT, 1e12)], [(0, LT, -1e12)], [(0, GT, -1e12), (0, LT, 1e12)],
                 [(0, EQ, float(int(v[12345])))], [(0, GT, -2147483647.0)], []):
        check(ctx, oracle, cols, [0], sels)


def test_int_extremes(ctx, oracle):
    v = np.array([-2**31, -2**31 + 1, -1, 0, 1, 2**31 - 2, 2**31 - 1] * 300, dtype=np.int64).astype(np.int32)
    cols = [RawColumn(DENSE_INT, 4, v, blocks_of(v.size, 1024))]
    for sels in ([(0, GT, -2147483648.0)], [(0, LT, 2147483647.0)], [(0, EQ, -2147483648.0)], [(0, EQ, 2147483647.0)],
                 [(0, GT, 2147483646.0)], [(0, LT, -2147483647.0)], [(0, GT, -2.0), (0, LT, 2.0)], [(0, GT, 5.0), (0, LT, 5.0)]):
        check(ctx, oracle, cols, [0], sels, proj=[0])


def test_tinyint_all_values_and_wrap(ctx, oracle):
    v = np.tile(np.arange(-128, 128, dtype=np.int16).astype(np.int8), 40)
    cols = [RawColumn(DENSE_TINYINT, 1, v, blocks_of(v.size, 1024))]
    for t in (200.0, 3e9, 127.0, 128.0, -128.0, -129.0, 256.0, 18.9, -5.5, float("nan")):
        for cond in (GT, LT, EQ):
            check(ctx, oracle, cols, [0], [(0, cond, t)])


# ---- strings: widths and IN-lists ----------------------------------------------------------------
@pytest.mark.parametrize("width", [1, 2, 3, 4, 5, 8, 9, 16])
def test_string_widths(ctx, oracle, width):
    rng = np.random.default_rng(width)
    n = 3000
    alphabet = np.frombuffer(b"ABCD", dtype=np.uint8)
    st = alphabet[rng.integers(0, 4, size=(n, width))]
    cols = [RawColumn(DENSE_STRING, width, st, blocks_of(n, 1024)), RawColumn(DENSE_INT, 4, np.arange(n, dtype=np.int32), blocks_of(n, 1024))]
    vals = [bytes(st[i]) for i in rng.integers(0, n, size=3)] + [b"A" * (width + 1), b""]
    check(ctx, oracle, cols, [0, 1], [(0, MATCH, vals)], proj=[1, 0])
    check(ctx, oracle, cols, [0, 1], [(0, MATCH, vals[:1]), (0, MATCH, vals[:2])], proj=[0])      # intersection
    check(ctx, oracle, cols, [0, 1], [(0, MATCH, [])], proj=[0])                                    # empty IN-list
    many = [bytes(st[i]) for i in range(12)]                                                        # > 8 values: device blob
    check(ctx, oracle, cols, [0, 1], [(0, MATCH, many)], proj=[0, 1])


# ---- more than 4 predicate columns: extra AND pass ---------------------------------------------------
def test_six_predicate_columns(ctx, oracle):
    rng = np.random.default_rng(99)
    n = 10_000
    br = blocks_of(n, 1024)
    cols = [RawColumn(DENSE_INT, 4, rng.integers(0, 100, size=n).astype(np.int32), br) for _ in range(3)]
    cols += [RawColumn(DENSE_TINYINT, 1, rng.integers(0, 100, size=n).astype(np.int8), br) for _ in range(3)]
    sels = [(i, GT, 10.0) for i in range(6)] + [(i, LT, 95.0) for i in range(6)]
    check(ctx, oracle, cols, list(range(6)), sels, proj=[5, 0])


# ---- errors mirror the reference's exceptions ---------------------------------------------------------
def test_errors(ctx):
    from immutable3_amd import native
    rng = np.random.default_rng(5)
    cols = make_cols(rng, 100, [100])
    seg = native.DeviceSegment(ctx, [c.native() for c in cols])
    for cond in (NOTMATCH, NOOP):
        with pytest.raises(native.Imm3Error) as e:
            native.DeviceQuery(ctx, seg, [1], [(0, cond, None)])
        assert e.value.code == native.ERR_UNSUPPORTED_CONDITION and "Unsupported condition" in e.value.msg
    for used, sel in (([2], (0, GT, 1.0)), ([2], (0, EQ, 1.0)), ([1], (0, MATCH, [b"CA"])), ([0], (0, MATCH, [b"CA"]))):
        with pytest.raises(native.Imm3Error) as e:
            native.DeviceQuery(ctx, seg, used, [sel])
        assert e.value.code == native.ERR_UNSUPPORTED_VECTOR and e.value.msg == "Unsupported column vector"
    with pytest.raises(native.Imm3Error) as e:
        native.DeviceQuery(ctx, seg, [0], [(3, GT, 1.0)])
    assert e.value.code == native.ERR_ARG
    seg.close()
    # zero batches: NotMatch still throws (iterator construction), a wrong vector type does not
    empty = native.DeviceSegment(ctx, [(DENSE_STRING, 2, np.zeros(0, np.uint8), 0, np.array([0], np.int32))])
    with pytest.raises(native.Imm3Error):
        native.DeviceQuery(ctx, empty, [0], [(0, NOTMATCH, [b"CA"])])
    q = native.DeviceQuery(ctx, empty, [0], [(0, GT, 1.0)])
    q.run()
    assert q.count() == 0 and q.n_batches == 0
    q.close()
    empty.close()
    # unknown codec id (CodecType has four values, Codec.scala:21-24)
    bad = native.DeviceSegment(ctx, [(7, 4, np.zeros(16, np.uint8), 16, np.array([0, 16], np.int32))])
    with pytest.raises(native.Imm3Error) as e:
        native.DeviceQuery(ctx, bad, [0], [])
    assert e.value.code == native.ERR_NO_CODEC
    bad.close()
    # misaligned columns: refused (the reference throws or mis-joins)
    a = RawColumn(DENSE_INT, 4, np.arange(100, dtype=np.int32), [100])
    b = RawColumn(DENSE_TINYINT, 1, np.arange(90, dtype=np.int8), [90])
    seg = native.DeviceSegment(ctx, [a.native(), b.native()])
    with pytest.raises(native.Imm3Error) as e:
        native.DeviceQuery(ctx, seg, [0, 1], [(1, GT, 1.0)])
    assert e.value.code == native.ERR_LAYOUT
    seg.close()


# ---- reservation too small: rows are re-gathered after growing ----------------------------------------
def test_reserve_too_small(ctx, oracle):
    rng = np.random.default_rng(11)
    cols = make_cols(rng, 20000, blocks_of(20000, 1024), small_ids=True)
    check(ctx, oracle, cols, [0, 1], [(0, GT, 0.0)], proj=[0, 1], limit=0, reserve=10)
    check(ctx, oracle, cols, [0, 1], [(0, GT, 0.0)], proj=[0, 1], limit=0, reserve=10**6)


# ---- query objects are re-runnable and independent -----------------------------------------------------
def test_rerun_and_two_queries(ctx, oracle):
    from immutable3_amd import native
    rng = np.random.default_rng(12)
    cols = make_cols(rng, 50000, blocks_of(50000, 1024), small_ids=True)
    seg = native.DeviceSegment(ctx, [c.native() for c in cols])
    q1 = native.DeviceQuery(ctx, seg, [0], [(0, GT, 0.0)], [0], 0)
    q2 = native.DeviceQuery(ctx, seg, [1, 0], [(0, LT, 0.0)], [1], 5)
    for _ in range(3):
        q1.run()
        q2.run()
    w1, c1 = oracle.scan_select([cols[0].ocol()], [(0, GT, 0.0)], 1024)
    w2, c2 = oracle.scan_select([cols[1].ocol(), cols[0].ocol()], [(0, LT, 0.0)], 1024)
    assert q1.count() == c1 and q1.bitmap().tolist() == w1.tolist()
    assert q2.count() == c2 and q2.bitmap().tolist() == w2.tolist()
    assert q1.row_count() == c1 and q2.row_count() == min(5, c2)
    q1.close(); q2.close(); seg.close()


# ---- survivor staging (projected column == predicate column): every selectivity incl. 0 % and 100 % ----------
@pytest.mark.parametrize("lo,hi", [(-1.0, 1e9), (1e9, 2e9), (0.0, 2.0), (49.0, 51.0), (10.0, 90.0)])
def test_staged_projection_selectivities(ctx, oracle, lo, hi):
    rng = np.random.default_rng(int(lo) + 7)
    n = 300_000 + 37
    ids = rng.integers(0, 100, size=n).astype(np.int32)
    age = rng.integers(0, 100, size=n).astype(np.int8)
    other = rng.integers(-5, 5, size=n).astype(np.int32)
    br = blocks_of(n, 1024)
    cols = [RawColumn(DENSE_INT, 4, ids, br), RawColumn(DENSE_TINYINT, 1, age, br), RawColumn(DENSE_INT, 4, other, br)]
    # both projected columns are predicate columns (staged); `other` is gathered the ordinary way
    check(ctx, oracle, cols, [1, 0, 2], [(0, GT, lo), (0, LT, hi), (1, GT, lo), (1, LT, hi)], proj=[1, 0, 2, 1])
    check(ctx, oracle, cols, [0], [(0, GT, lo), (0, LT, hi)], proj=[0])
    check(ctx, oracle, cols, [1], [(0, GT, lo), (0, LT, hi)], proj=[0])


# ---- dense survivor lists: more survivors per 16-tile span than one LDS list batch holds (4096) ------------------
@pytest.mark.parametrize("keep", [1.0, 0.6, 0.26, 0.24])
def test_projection_at_high_selectivity(ctx, oracle, keep):
    rng = np.random.default_rng(int(keep * 100))
    n = 16384 * 3 + 5000
    cols = make_cols(rng, n, blocks_of(n, 1024), small_ids=True)           # ids uniform in [-50, 50)
    t = -50 + 100 * (1 - keep)
    check(ctx, oracle, cols, [0, 1, 2], [(0, GT, float(t) - 0.5)], proj=[2, 0, 1])
    check(ctx, oracle, cols, [0], [(0, GT, float(t) - 0.5)], proj=[0], limit=9000)


# ---- survivor records (k_filter_tile STAGE -> k_emit): every record size at every fill level ----
STAGED_SHAPES = [
    # predicate columns (indices into [a:i32, b:i32, c:i8, d:i8, s:s2]); all of them are projected, plus `extra` gathered columns
    ([0], []), ([2], []), ([4], []), ([0, 2], []), ([0, 1], []), ([0, 1, 2], []), ([0, 2, 3], []), ([2, 3, 4], []), ([0, 2, 4], []),
    ([0, 1, 4], []), ([2], [0, 4]), ([0, 1], [2, 4]), ([], [0, 2]),
]


@pytest.mark.parametrize("pred_cols,extra", STAGED_SHAPES)
def test_staged_projection_every_record_size_and_fill(pred_cols, extra):
    """1-, 2- and 4-dword records from tiles with 0, a few, most and all rows surviving (a 4-dword record tile with more
    than 512 survivors does not fit the wave's LDS buffer: round 2 found that case writing past it)."""
    from immutable3_amd import native, synth
    ctx = native.Context(0)
    n = 300_000 + 77
    br = blocks_of(n, 1024)
    rng = np.random.default_rng(len(pred_cols) * 10 + len(extra))
    a = synth.uniform_int30(21, n)
    b = synth.uniform_int30(22, n)
    c = synth.uniform_below(23, n, 100, np.int8)
    d = (synth.uniform_below(24, n, 100, np.int8) - 50).astype(np.int8)
    s = synth.state_codes(25, n)
    data = [a, b, c, d, s]
    cols = [RawColumn(DENSE_INT, 4, a, br), RawColumn(DENSE_INT, 4, b, br), RawColumn(2, 1, c, br), RawColumn(2, 1, d, br), RawColumn(3, 2, s, br)]
    seg = native.DeviceSegment(ctx, [x.native() for x in cols])
    # per fill level, the threshold each numeric predicate uses: keep = value > t
    levels = {"none": {0: 2.0 ** 31, 1: 2.0 ** 31, 2: 127.0, 3: 127.0}, "few": {0: 0.97 * 2 ** 30, 1: 0.9 * 2 ** 30, 2: 95.0, 3: 45.0},
              "most": {0: 0.1 * 2 ** 30, 1: 0.05 * 2 ** 30, 2: 5.0, 3: -45.0}, "all": {0: -1.0, 1: -1.0, 2: -1.0, 3: -128.5}}
    codes = {"none": [b"??"], "few": [b"CA"], "most": [bytes(x) for x in np.unique(s, axis=0)[:40]], "all": [bytes(x) for x in np.unique(s, axis=0)]}
    used = sorted(set(pred_cols) | set(extra))
    pos = {u: i for i, u in enumerate(used)}
    for level in ("none", "few", "most", "all"):
        sels, keep = [], np.ones(n, bool)
        for pc in pred_cols:
            if pc == 4:
                if len(codes[level]) > 8:        # (long IN-lists go through the generic kernel: not a staged shape)
                    lst = codes[level][:8]
                else:
                    lst = codes[level]
                sels.append((pos[pc], native.MATCH, lst))
                m = np.zeros(n, bool)
                for v in lst:
                    m |= (s[:, 0] == v[0]) & (s[:, 1] == v[1])
                keep &= m
            else:
                sels.append((pos[pc], GT, float(levels[level][pc])))
                keep &= data[pc] > levels[level][pc]
        q = native.DeviceQuery(ctx, seg, used, sels, list(range(len(used))), 0)
        for _ in range(2):                       # twice: the second run re-uses arenas and tables
            q.run()
        rows = np.flatnonzero(keep)
        assert q.count() == rows.size, (level, pred_cols)
        idx, vals = q.fetch_rows()
        assert idx.size == rows.size and (idx == rows).all(), (level, pred_cols)
        for j, u in enumerate(used):
            if u == 4:
                assert (vals[j].reshape(-1, 2) == s[rows]).all(), (level, pred_cols, u)
            else:
                assert (vals[j].view("<i4" if u < 2 else np.int8).reshape(-1) == data[u][rows]).all(), (level, pred_cols, u)
        q.close()
    seg.close()
    ctx.close()


# ---- count-only runs (imm3_query_run_count): the same count, no bitmap ------------------------------------------------------
def test_count_only_runs_match_the_bitmap_path(ctx, oracle):
    from immutable3_amd import native
    rng = np.random.default_rng(99)
    n = 300_000 + 123
    br = blocks_of(n, 1024)
    cols = make_cols(rng, n, br, small_ids=True)
    seg = native.DeviceSegment(ctx, [c.native() for c in cols])
    shapes = [([0], [(0, GT, -10.0), (0, LT, 25.0)]), ([1], [(0, GT, 18.0), (0, LT, 30.0)]), ([2], [(0, MATCH, [b"CA", b"TX"])]),
              ([0, 1], [(0, GT, 0.0), (1, LT, 64.0)]), ([1, 2, 0], [(0, GT, -5.0), (1, MATCH, [b"NY"]), (2, LT, 40.0)]), ([0], []),
              ([1], [(0, GT, 500.0)]),                                        # always false after narrowing: d2b(500) = -12 ... not empty; see below
              ([0], [(0, GT, 10.0), (0, LT, 5.0)])]                           # empty interval: answered by memset
    for used, sels in shapes:
        q = native.DeviceQuery(ctx, seg, used, sels)
        q.run_select()
        want, words = q.count(), q.bitmap()
        ow, oc = oracle.scan_select([cols[i].ocol() for i in used], sels, 1024, 1)
        assert want == oc and words.tolist() == ow.tolist()
        q.run_count()
        assert q.count() == want
        single_launch = len(used) <= 3 and want != 0 or not sels
        try:
            got = q.bitmap()
            assert got.tolist() == ow.tolist()                                  # (a chain that is not one launch keeps its bitmap)
        except native.Imm3Error as e:
            assert e.code == native.ERR_STATE and "count-only" in e.msg
        q.run_select()                                                          # a full run brings the bitmap back
        assert q.bitmap().tolist() == ow.tolist() and q.count() == want
        q.close()
    # six predicate columns = two tile passes: run_count falls back to the full select
    cols6 = [RawColumn(DENSE_INT, 4, rng.integers(-50, 50, size=n).astype(np.int32), br) for _ in range(6)]
    seg6 = native.DeviceSegment(ctx, [c.native() for c in cols6])
    sels = [(c, GT, -20.0) for c in range(6)]
    q = native.DeviceQuery(ctx, seg6, list(range(6)), sels)
    q.run_count()
    ow, oc = oracle.scan_select([c.ocol() for c in cols6], sels, 1024, 1)
    assert q.count() == oc and q.bitmap().tolist() == ow.tolist()
    q.close()
    seg6.close()
    seg.close()
