"""BASELINE.json's full sizes (100 M rows) on the GPU, checked through size-independent properties
(the oracle is only run on a bounded sample): popcount(bitmap) == count == closed-form count,
complement predicates partition the rows, AND of single-column filters == fused filter, projected
rows are sorted, satisfy the predicate and equal a numpy evaluation on sampled tiles."""
import numpy as np
import pytest

from conftest import DENSE_INT, DENSE_STRING, DENSE_TINYINT, EQ, GT, LT, MATCH, RawColumn, blocks_of

pytestmark = pytest.mark.gpu
N = 100_000_000


@pytest.fixture(scope="module")
def big():
    from immutable3_amd import native, synth
    assert native.device_count() >= 1
    ctx = native.Context(0)
    ids = np.arange(N, dtype=np.int32)
    age = synth.uniform_below(2, N, 100, np.int8)
    st = synth.state_codes(3, N)
    offs = lambda w: synth.block_offsets(N, w)
    seg = native.DeviceSegment(ctx, [(DENSE_INT, 4, ids.view(np.uint8), N * 4, offs(4)),
                                     (DENSE_TINYINT, 1, age.view(np.uint8), N, offs(1)),
                                     (DENSE_STRING, 2, st.reshape(-1), N * 2, offs(2))])
    yield ctx, seg, ids, age, st
    seg.close()
    ctx.close()


def popcount(words):
    return int(np.unpackbits(words.view(np.uint8)).sum())


def test_c3_range_and_project_100m(big, oracle):
    from immutable3_amd import native
    ctx, seg, ids, age, st = big
    sels = [(0, GT, 18.0), (0, LT, 30.0), (1, GT, 1e6), (1, LT, 9e7)]
    q = native.DeviceQuery(ctx, seg, [1, 0], sels, [1, 0], 0)     # used = [age, id]; project (id, age)
    assert q.plan()["single_pass"], q.plan()                       # BASELINE's C3 at its full size: ONE launch (the cost model's choice from the sample)
    q.run()
    count = q.count()
    words = q.bitmap()
    assert q.n_batches == 97657 and q.total_words == (N + 63) // 64
    keep = (age > 18) & (age < 30) & (ids > 1_000_000) & (ids < 90_000_000)
    assert count == int(keep.sum()) == popcount(words)
    assert words.tobytes() == np.packbits(keep, bitorder="little").tobytes()
    idx, vals = q.fetch_rows()
    assert idx.shape[0] == count
    assert (np.diff(idx.astype(np.int64)) > 0).all()                # ascending row order
    expect_rows = np.flatnonzero(keep)
    assert (idx == expect_rows).all()
    assert (vals[0].view("<i4").reshape(-1) == ids[expect_rows]).all()
    assert (vals[1].view(np.int8).reshape(-1) == age[expect_rows]).all()
    q.close()
    # oracle on a bounded sample: the first 2M rows as their own segment
    m = 2_000_000
    cols = [RawColumn(DENSE_TINYINT, 1, age[:m], blocks_of(m, 1024)), RawColumn(DENSE_INT, 4, ids[:m], blocks_of(m, 1024))]
    ow, oc = oracle.scan_select([c.ocol() for c in cols], sels, 1024, 1)
    assert words[: ow.size].tolist() == ow.tolist()


def test_partition_and_fusion_100m(big):
    from immutable3_amd import native
    ctx, seg, ids, age, st = big
    def run(used, sels):
        q = native.DeviceQuery(ctx, seg, used, sels)
        q.run()
        out = (q.bitmap(), q.count())
        q.close()
        return out
    w_lt, c_lt = run([0], [(0, LT, 5e7)])
    w_ge, c_ge = run([0], [(0, GT, 5e7 - 1)])
    assert c_lt + c_ge == N and c_lt == 50_000_000
    tail = np.uint64((1 << (N % 64)) - 1) if N % 64 else np.uint64(0xFFFFFFFFFFFFFFFF)
    full = np.full(w_lt.size, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    full[-1] = tail
    assert ((w_lt ^ w_ge) == full).all() and not (w_lt & w_ge).any()
    w_age, _ = run([1], [(0, GT, 18.0), (0, LT, 30.0)])
    w_both, c_both = run([1, 0], [(0, GT, 18.0), (0, LT, 30.0), (1, LT, 5e7)])
    assert ((w_age & w_lt) == w_both).all() and popcount(w_both) == c_both
    w_none, c_none = run([0], [])
    assert c_none == N and (w_none == full).all()


def test_narrow_only_chains_100m(big):
    """The in-lane instances of the scan + select kernel (int8 / 2-byte strings only: csrc/imm3_kernels.hip, lane_tile) at the full
    size, against numpy: the whole bitmap, the count, the count-only run, complements and the AND of single-column filters."""
    from immutable3_amd import native
    ctx, seg, ids, age, st = big
    codes = st.view("<u2").reshape(-1)
    ca, ny = int.from_bytes(b"CA", "little"), int.from_bytes(b"NY", "little")
    def run(used, sels):
        q = native.DeviceQuery(ctx, seg, used, sels)
        q.run()
        w, c = q.bitmap(), q.count()
        q.run_count()
        assert q.count() == c
        q.close()
        return w, c
    def words(keep):
        return np.packbits(keep, bitorder="little").view("<u8")           # (N is a multiple of 64)
    k_age = (age > 18) & (age < 30)
    k_st = (codes == ca) | (codes == ny)
    w_age, c_age = run([1], [(0, GT, 18.0), (0, LT, 30.0)])
    assert c_age == int(k_age.sum()) and (w_age == words(k_age)).all()
    w_st, c_st = run([2], [(0, MATCH, [b"CA", b"NY"])])
    assert c_st == int(k_st.sum()) and (w_st == words(k_st)).all()
    w_both, c_both = run([1, 2], [(0, GT, 18.0), (0, LT, 30.0), (1, MATCH, [b"CA", b"NY"])])
    assert ((w_age & w_st) == w_both).all() and c_both == popcount(w_both) == int((k_age & k_st).sum())
    w_lo, c_lo = run([1], [(0, LT, 19.0)])
    w_hi, c_hi = run([1], [(0, GT, 18.0)])
    assert c_lo + c_hi == N and not (w_lo & w_hi).any() and ((w_lo | w_hi) == np.uint64(0xFFFFFFFFFFFFFFFF)).all()


def test_c4_match_project_100m(big):
    from immutable3_amd import native
    ctx, seg, ids, age, st = big
    q = native.DeviceQuery(ctx, seg, [2, 0, 1], [(0, MATCH, [b"CA"])], [1, 0, 2], 0)   # project (id, state, age)
    q.run()
    keep = (st[:, 0] == ord("C")) & (st[:, 1] == ord("A"))
    assert q.count() == int(keep.sum())
    idx, vals = q.fetch_rows()
    rows = np.flatnonzero(keep)
    assert (idx == rows).all()
    assert (vals[0].view("<i4").reshape(-1) == ids[rows]).all()
    assert (vals[1] == st[rows]).all()
    assert (vals[2].view(np.int8).reshape(-1) == age[rows]).all()
    q.close()
    q = native.DeviceQuery(ctx, seg, [2, 0, 1], [(0, MATCH, [b"CA"])], [1, 0, 2], 10)
    q.run()
    idx10, v10 = q.fetch_rows()
    assert (idx10 == rows[:10]).all() and (v10[0].view("<i4").reshape(-1) == ids[rows[:10]]).all()
    q.close()


def test_maximum_segment_size():
    """The largest DENSE_INT segment the reference format can describe: blockOffset is an Array[Int]
    (core/storage/Segment.scala:33), so a column holds < 2 GiB = 536 870 911 int32 rows.  Closed-form counts, the tail
    of the segment (partial last tile, row indices near 2^29) and a mid-segment window."""
    from immutable3_amd import native, synth
    n = (2 ** 31 - 1) // 4
    ctx = native.Context(0)
    ids = np.arange(n, dtype=np.int32)
    offs = synth.block_offsets(n, 4)
    assert int(offs[-1]) == n * 4 and offs.dtype == np.int32
    seg = native.DeviceSegment(ctx, [(DENSE_INT, 4, ids.view(np.uint8), n * 4, offs)])
    del ids
    q = native.DeviceQuery(ctx, seg, [0], [(0, GT, 1000.0), (0, LT, float(n - 1000))])
    q.run()
    assert q.count() == n - 2001
    w = q.bitmap()
    assert w.size == (n + 63) // 64 and popcount(w) == n - 2001
    assert int(w[0]) == 0 and int(w[-1]) == 0 and int(w[w.size // 2]) == 2 ** 64 - 1
    q.close()
    q = native.DeviceQuery(ctx, seg, [0], [(0, GT, float(n - 70))], [0], 0)
    q.run()
    idx, vals = q.fetch_rows()
    assert idx.tolist() == list(range(n - 69, n)) and vals[0].view("<i4").reshape(-1).tolist() == list(range(n - 69, n))
    q.close()
    q = native.DeviceQuery(ctx, seg, [0], [(0, GT, 2.0 ** 28 - 3), (0, LT, 2.0 ** 28 + 3)], [0], 3)
    q.run()
    idx, vals = q.fetch_rows()
    assert idx.tolist() == [2 ** 28 - 2, 2 ** 28 - 1, 2 ** 28]
    q.close()
    seg.close()
    ctx.close()


def test_group_by_state_100m(big):
    """Group-by aggregation at the bench's size (k_group_agg_lanes), all rows and under a range predicate, against a numpy
    evaluation: group keys in first-seen order, counts, max / min of the int8 column, first rows."""
    from immutable3_amd import native
    ctx, seg, ids, age, st = big
    code = st.view("<u2").reshape(-1)                           # the two bytes of a state, little-endian
    for sels, keep in (([], None), ([(1, GT, 18.0), (1, LT, 30.0)], (age > 18) & (age < 30))):
        for kind, red in ((native.AGG_MAX, np.maximum), (native.AGG_MIN, np.minimum)):
            q = native.DeviceQuery(ctx, seg, [0, 1, 2], sels, (), 0, 1024, group_cols=[2], aggs=[(native.AGG_COUNT, 0), (kind, 1)])
            q.run()
            keys, first, counts, vals = q.fetch_groups()
            q.close()
            rows = np.arange(N) if keep is None else np.flatnonzero(keep)
            c, a = code[rows], age[rows]
            uniq, first_pos = np.unique(c, return_index=True)
            order = np.argsort(first_pos)                       # first-seen order
            want_keys = uniq[order]
            assert keys.astype(np.uint64).tolist() == want_keys.astype(np.uint64).tolist()
            assert first.tolist() == rows[first_pos[order]].tolist()
            cnt = np.bincount(c, minlength=65536)
            assert counts.tolist() == cnt[want_keys].tolist() and int(counts.sum()) == rows.size
            ext = np.full(65536, -128 if kind == native.AGG_MAX else 127, np.int8)
            red.at(ext, c, a)
            assert vals[:, 1].tolist() == ext[want_keys].astype(np.int64).tolist()


@pytest.mark.parametrize("pred_cols", [[2], [0, 2], [0, 1], [0, 1, 2]])
def test_staged_projection_many_tiles_per_wave(pred_cols):
    """12.6 M rows = 12 305 tiles: every wave of the staging launch takes 4+ tiles, so its LDS record buffer wraps, flushes and (for
    4-dword records at high fill) hands whole tiles straight to the arena several times -- at sparse, dense and full tiles."""
    from immutable3_amd import native, synth
    ctx = native.Context(0)
    n = 12_600_000 + 333
    offs = lambda w: synth.block_offsets(n, w)
    a = synth.uniform_int30(31, n)
    b = synth.uniform_int30(32, n)
    c = synth.uniform_below(33, n, 100, np.int8)
    data = [a, b, c]
    seg = native.DeviceSegment(ctx, [(DENSE_INT, 4, a.view(np.uint8), n * 4, offs(4)), (DENSE_INT, 4, b.view(np.uint8), n * 4, offs(4)),
                                     (DENSE_TINYINT, 1, c.view(np.uint8), n, offs(1))])
    levels = {"few": {0: 0.97 * 2 ** 30, 1: 0.9 * 2 ** 30, 2: 95.0}, "most": {0: 0.1 * 2 ** 30, 1: 0.05 * 2 ** 30, 2: 5.0}, "all": {0: -1.0, 1: -1.0, 2: -1.0}}
    pos = {u: i for i, u in enumerate(pred_cols)}
    for level in ("few", "most", "all"):
        sels = [(pos[pc], GT, float(levels[level][pc])) for pc in pred_cols]
        keep = np.ones(n, bool)
        for pc in pred_cols:
            keep &= data[pc] > levels[level][pc]
        q = native.DeviceQuery(ctx, seg, pred_cols, sels, list(range(len(pred_cols))), 0)
        q.run()
        q.run()
        rows = np.flatnonzero(keep)
        assert q.count() == rows.size
        idx, vals = q.fetch_rows()
        assert idx.size == rows.size and (idx == rows).all(), (level, pred_cols)
        for j, u in enumerate(pred_cols):
            assert (vals[j].view("<i4" if u < 2 else np.int8).reshape(-1) == data[u][rows]).all(), (level, pred_cols, u)
        q.close()
    seg.close()
    ctx.close()
