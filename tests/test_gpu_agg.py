"""GPU suite: group-by aggregation (ProjectAggOp / ProjectAggregateQueueOp) through the C ABI and
the operator mirror, against the oracle's restatement (oracle_np.project_agg / combine_agg)."""
import os

import numpy as np
import pytest

from conftest import DENSE_INT, DENSE_STRING, DENSE_TINYINT, EQ, GT, LT, MATCH, RawColumn, blocks_of
from immutable3_amd import Count, GT as QGT, LT as QLT, And, Match, Max, Min, NoSelect, ProjectAgg, Query, Select, Sum
from immutable3_amd import native, synth
from oracle import oracle_c, oracle_np

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KIND = {"count": native.AGG_COUNT, "min": native.AGG_MIN, "max": native.AGG_MAX}
CODES = [b"CA", b"NY", b"TX", b"WA", b"VA", b"DC", b"CT"]


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0)
    yield c
    c.close()


def gpu_groups(ctx, cols, used, sels, group, aggs, block_size=1024):
    seg = native.DeviceSegment(ctx, [c.native() for c in cols])
    q = native.DeviceQuery(ctx, seg, used, sels, (), 0, block_size, group_cols=group, aggs=[(KIND[k], c) for k, c in aggs])
    q.run()
    keys, first, counts, vals = q.fetch_groups()
    q.close()
    seg.close()
    return keys, first, counts, vals


def check(ctx, cols, used, sels, group, aggs, block_size=1024):
    ucols = [cols[i] for i in used]
    _, _, masks = oracle_np.scan_select([c.npcol() for c in ucols], sels, block_size)
    expect = oracle_np.project_agg([c.npcol() for c in ucols], group, aggs, masks)
    if all(hasattr(c, "ocol") for c in ucols):      # ... and the C twin: two restatements must agree before they grade the GPU
        words, _ = oracle_c.scan_select([c.ocol() for c in ucols], sels, block_size)
        twin = oracle_c.project_agg([c.ocol() for c in ucols], group, aggs, words)
        assert list(twin.items()) == list(expect.items())
    keys, first, counts, vals = gpu_groups(ctx, cols, used, sels, group, aggs, block_size)
    assert keys.shape[0] == len(expect)
    got = []
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
            c = ucols[ci]
            if kind == "count":
                st.append(int(counts[g]))
            elif getattr(c, "_dense", c).codec == DENSE_STRING:
                st.append(int(vals[g, j]).to_bytes(8, "big", signed=True)[8 - c.width:].decode())
            else:
                st.append(float(int(vals[g, j])))
        got.append(("_".join(parts), st))
    assert got == [(k, v) for k, v in expect.items()]          # same groups, same first-seen order, same values
    assert (np.diff(first.astype(np.int64)) > 0).all()
    assert int(counts.sum()) == sum(int(m.sum()) for m in masks)
    return got


def make_cols(rng, n, block_rows, id_range=50):
    ids = rng.integers(-id_range, id_range, size=n).astype(np.int32)
    age = rng.integers(-128, 128, size=n).astype(np.int8)
    st = np.array([list(CODES[i]) for i in rng.integers(0, len(CODES), size=n)], dtype=np.uint8).reshape(n, 2)
    return [RawColumn(DENSE_INT, 4, ids, block_rows), RawColumn(DENSE_TINYINT, 1, age, block_rows), RawColumn(DENSE_STRING, 2, st, block_rows)]


def test_test100_group_by_state_kat(ctx):
    """Closed form on the C1 table: state = CODES7[i % 7] -> 15,15,14,14,14,14,14 rows; first-seen order CA..CT."""
    t = synth.test_100()
    cols = [RawColumn(DENSE_INT, 4, t["id"], [100]), RawColumn(DENSE_STRING, 2, t["state"], [100]), RawColumn(DENSE_TINYINT, 1, t["age"], [100])]
    got = check(ctx, cols, [0, 1, 2], [], [1], [("count", 0), ("max", 2), ("min", 2), ("max", 0)])
    assert [k for k, _ in got] == synth.CODES7
    assert [v[0] for _, v in got] == [15, 15, 14, 14, 14, 14, 14]
    age = t["age"].astype(int)
    assert [v[1] for _, v in got] == [float(age[k::7].max()) for k in range(7)]
    assert [v[3] for _, v in got] == [float(99 - ((99 - k) % 7)) for k in range(7)]
    # select max(age) from test_100 where (age > 18 and age < 30) -> single group "" (no group by)
    got = check(ctx, cols, [2], [(0, GT, 18.0), (0, LT, 30.0)], [], [("max", 0), ("count", 0)])
    assert got == [("", [29.0, 11])]


LAYOUTS = [(0, []), (1, [1]), (100, [100]), (1025, [1024, 1]), (25, [4, 4, 1, 4, 4, 1, 4, 3]), (5000, blocks_of(5000, 1024)),
           (70000, blocks_of(70000, 1024)), (130, [64, 0, 66])]


@pytest.mark.parametrize("n,block_rows", LAYOUTS)
def test_random_group_queries(ctx, n, block_rows):
    rng = np.random.default_rng(77 + n)
    for trial in range(4):
        cols = make_cols(rng, n, block_rows)
        used = [0, 1, 2]
        sels = [[], [(1, GT, 0.0)], [(2, MATCH, [b"CA", b"NY", b"TX"])], [(0, GT, -10.0), (0, LT, 10.0)]][trial]
        group = [[2], [1, 2], [], [0]][trial]
        aggs = [[("count", 0), ("max", 1)], [("min", 0), ("max", 2), ("count", 2)], [("max", 0), ("min", 1)], [("count", 0), ("min", 1), ("max", 1), ("max", 2)]][trial]
        check(ctx, cols, used, sels, group, aggs)


def test_high_cardinality_groups_spill_past_lds(ctx):
    """200k rows, ~100k distinct int keys: the per-work-group LDS tables overflow and rows go to the global table."""
    rng = np.random.default_rng(5)
    n = 200_000
    ids = rng.integers(0, 100_000, size=n).astype(np.int32)
    age = rng.integers(0, 100, size=n).astype(np.int8)
    cols = [RawColumn(DENSE_INT, 4, ids, blocks_of(n, 1024)), RawColumn(DENSE_TINYINT, 1, age, blocks_of(n, 1024))]
    check(ctx, cols, [0, 1], [], [0], [("count", 0), ("max", 1), ("min", 1)])
    # two-column key of 5 bytes incl. negative ints
    ids2 = rng.integers(-2**31, 2**31, size=3000, dtype=np.int64).astype(np.int32)
    ids2[::3] = -1
    cols = [RawColumn(DENSE_INT, 4, ids2, blocks_of(3000, 1024)), RawColumn(DENSE_TINYINT, 1, age[:3000], blocks_of(3000, 1024))]
    check(ctx, cols, [0, 1], [(1, GT, 10.0)], [0, 1], [("count", 0)])


def test_all_ones_key_and_extremes(ctx):
    """An 8-byte key of all 0xFF bytes (two int columns both -1) uses the table's reserved slot."""
    a = np.array([-1, -1, 5, -1, 5, -2**31, 2**31 - 1] * 50, dtype=np.int64).astype(np.int32)
    b = np.array([-1, 7, -1, -1, -1, 0, 0] * 50, dtype=np.int64).astype(np.int32)
    cols = [RawColumn(DENSE_INT, 4, a, blocks_of(a.size, 64)), RawColumn(DENSE_INT, 4, b, blocks_of(a.size, 64))]
    got = check(ctx, cols, [0, 1], [], [0, 1], [("count", 0), ("min", 0), ("max", 1)], block_size=64)
    assert got[0] == ("-1_-1", [100, -1.0, -1.0])


def test_agg_errors(ctx):
    rng = np.random.default_rng(1)
    cols = make_cols(rng, 100, [100])
    seg = native.DeviceSegment(ctx, [c.native() for c in cols])
    with pytest.raises(native.Imm3Error) as e:      # MIN over a String vector: "bad aggregator for this data type"
        native.DeviceQuery(ctx, seg, [2], [], (), 0, 1024, group_cols=[], aggs=[(native.AGG_MIN, 0)])
    assert e.value.code == native.ERR_UNSUPPORTED_VECTOR and "bad aggregator" in e.value.msg
    with pytest.raises(native.Imm3Error):           # unknown aggregate kind
        native.DeviceQuery(ctx, seg, [0], [], (), 0, 1024, group_cols=[], aggs=[(9, 0)])
    wide = [RawColumn(DENSE_INT, 4, np.arange(10, dtype=np.int32), [10]) for _ in range(3)]
    seg2 = native.DeviceSegment(ctx, [c.native() for c in wide])
    with pytest.raises(native.Imm3Error) as e:      # 12-byte key
        native.DeviceQuery(ctx, seg2, [0, 1, 2], [], (), 0, 1024, group_cols=[0, 1, 2], aggs=[(native.AGG_COUNT, 0)])
    assert e.value.code == native.ERR_ARG
    seg.close(); seg2.close()


def test_engine_agg_over_segments_and_double_repr():
    """Engine.execute on ProjectAgg over the 3-segment quirk_25 table: per-segment ProjectAggOp + combine by key."""
    from immutable3_amd.operators import Engine, GpuSegmentManager
    from immutable3_amd.storage import SegmentManager
    g = GpuSegmentManager(SegmentManager(GOLDEN))
    q = Query("quirk_25", Select("id", QGT(2)), ProjectAgg([Count("id"), Max("id"), Min("age"), Max("state")], ["state"]))
    res = Engine(g).execute_agg(q)
    ids = np.arange(25)
    states = [synth.CODES7[i % 7] for i in range(25)]
    ages = (ids * 5) % 11 - 5
    keep = ids > 2
    order = []
    for i in np.flatnonzero(keep):
        if states[i] not in order:
            order.append(states[i])
    assert list(res) == order
    for s in order:
        sel = [i for i in np.flatnonzero(keep) if states[i] == s]
        a = res[s]
        assert a["id_count"].repr() == str(len(sel))
        assert a["id_max"].repr() == f"{max(sel)}.0"
        assert a["age_min"].repr() == f"{int(min(ages[sel]))}.0"
        assert a["state_max"].repr() == s
    rows = [tuple(r) for r in Engine(g).execute(q)]
    assert rows[0] == (res[order[0]]["id_count"].repr(), res[order[0]]["id_max"].repr(), res[order[0]]["age_min"].repr(), order[0])
    with pytest.raises(Exception, match="Unknown Aggregate type"):
        list(Engine(g).execute(Query("quirk_25", NoSelect, ProjectAgg([Sum("id")], []))))
    # Min over a STRING column is mapped to MaxStringAggr by the reference's planner (Engine.scala:145)
    res = Engine(g).execute_agg(Query("quirk_25", NoSelect, ProjectAgg([Min("state")], [])))
    assert res[""]["state_min"].repr() == max(states)
    g.close()
    from immutable3_amd.operators import java_double_to_string
    assert [java_double_to_string(v) for v in (0.0, 89.0, -5.0, 9999999.0, 1e7, 12345678.0, 2147483647.0, -2147483648.0)] == \
        ["0.0", "89.0", "-5.0", "9999999.0", "1.0E7", "1.2345678E7", "2.147483647E9", "-2.147483648E9"]


# ---- k_group_agg_lanes (key <= 2 bytes, <= 63 keys, <= 1 min/max aggregate): every key shape x value width ----
def _codes(rng, n, k, first_bytes=None):
    """n two-byte codes drawn from k distinct ones; `first_bytes` distinct first characters."""
    first_bytes = first_bytes or k
    pool = np.array([[65 + (i % first_bytes), 97 + (i // first_bytes)] for i in range(k)], dtype=np.uint8)
    return pool[rng.integers(0, k, size=n)]


LANES_CASES = [
    # (key columns, aggregates) over columns [id i32, age i8, state s2, flag i8, code s2]
    ([2], [("count", 0)]),                                   # KS 1, VW 0
    ([2], [("count", 0), ("max", 1)]),                       # KS 1, VW 1 (the bench query)
    ([2], [("min", 1), ("count", 1), ("count", 0)]),         # KS 1, VW 1, MIN over signed bytes
    ([2], [("max", 4)]),                                     # KS 1, VW 2: max over a string column
    ([2], [("count", 2), ("min", 0)]),                       # KS 1, VW 4
    ([2], [("max", 0)]),                                     # KS 1, VW 4
    ([3], [("count", 0), ("max", 1)]),                       # KS 0
    ([3], [("min", 0)]),                                     # KS 0, VW 4
    ([3, 1], [("count", 0), ("max", 4)]),                    # KS 2 (two byte columns), VW 2
    ([1, 3], [("min", 1)]),                                  # KS 2, other column order
    ([2], [("max", 1), ("min", 1), ("count", 0)]),           # two value aggregates over the same 1-byte column
    ([2], [("min", 3), ("count", 2), ("max", 1)]),           # ... over two 1-byte columns
    ([3], [("max", 1), ("min", 3)]),                         # ... KS 0
    ([3, 1], [("min", 1), ("max", 1)]),                      # ... KS 2
    ([2], [("max", 0), ("max", 1)]),                         # 4 + 1 bytes: not a lanes shape (k_group_agg_direct), same answers
]


def _lanes_cols(rng, n, n_codes=51, age_values=4, first_bytes=19):
    br = blocks_of(n, 1024)
    ids = rng.integers(-2**31, 2**31, size=n, dtype=np.int64).astype(np.int32)
    age = (rng.integers(0, age_values, size=n) * 37 - 60).astype(np.int8)       # few distinct values, both signs
    st = _codes(rng, n, n_codes, first_bytes)
    flag = rng.integers(-3, 4, size=n).astype(np.int8)                            # 7 values
    code = _codes(rng, n, 40)
    return [RawColumn(DENSE_INT, 4, ids, br), RawColumn(DENSE_TINYINT, 1, age, br), RawColumn(DENSE_STRING, 2, st, br),
            RawColumn(DENSE_TINYINT, 1, flag, br), RawColumn(DENSE_STRING, 2, code, br)]


@pytest.mark.parametrize("group,aggs", LANES_CASES)
def test_lanes_form_every_shape(ctx, group, aggs):
    rng = np.random.default_rng(len(group) * 100 + len(aggs))
    cols = _lanes_cols(rng, 300_000)
    used = [0, 1, 2, 3, 4]
    check(ctx, cols, used, [], group, aggs)                                       # every row
    check(ctx, cols, used, [(0, GT, 0.0), (3, LT, 2.0)], group, aggs)             # a selection: rows that are not selected go to the trash slot
    check(ctx, cols, used, [(0, GT, 2.0 ** 31 - 4096)], group, aggs)              # nearly nothing selected: most tiles skipped


def test_lanes_form_pipeline_depth_and_rare_keys(ctx):
    """7.2 M rows = 7032 tiles: more than 3 tiles per wave (the prefetch ring wraps), plus keys that appear once, late."""
    rng = np.random.default_rng(99)
    n = 7_200_000
    br = blocks_of(n, 1024)
    st = _codes(rng, n, 51, 19)
    st[n - 5] = [90, 90]                                                          # "ZZ" once, in the last tile
    st[3_333_333] = [90, 65]                                                      # "ZA" once, mid-way
    age = rng.integers(-128, 128, size=n).astype(np.int8)
    ids = np.arange(n, dtype=np.int32)
    cols = [RawColumn(DENSE_INT, 4, ids, br), RawColumn(DENSE_STRING, 2, st, br), RawColumn(DENSE_TINYINT, 1, age, br)]
    got = check(ctx, cols, [0, 1, 2], [], [1], [("count", 0), ("max", 2)])
    assert [k for k, _ in got][-2:] == ["ZA", "ZZ"] and got[-1][1][0] == 1
    check(ctx, cols, [0, 1, 2], [(2, GT, 18.0), (2, LT, 30.0)], [1], [("count", 0), ("min", 2)])


def test_lanes_form_overflows_into_the_next_form(ctx):
    """More than 63 keys, or more than 29 distinct first key bytes: overflow = 3, the host re-runs k_group_agg_direct
    (and the general kernel after that when even 254 slots do not do); a second run of the same query starts there."""
    rng = np.random.default_rng(7)
    n = 200_000
    br = blocks_of(n, 1024)
    ids = rng.integers(-1000, 1000, size=n).astype(np.int32)
    byte_key = rng.integers(-100, 100, size=n).astype(np.int8)                    # 200 keys of one byte
    many = _codes(rng, n, 64, 8)                                                  # 64 keys: one too many
    pages = _codes(rng, n, 40, 40)                                                # 40 keys, 40 distinct first bytes
    wide = _codes(rng, n, 600, 25)                                                # 600 keys: past the direct form as well
    mid = rng.integers(-50, 50, size=n).astype(np.int8)                            # 100 keys: the 127-key instance of the lanes form
    for key in (byte_key, mid, many, pages, wide):
        kc = RawColumn(DENSE_TINYINT, 1, key, br) if key.dtype == np.int8 else RawColumn(DENSE_STRING, 2, key, br)
        cols = [RawColumn(DENSE_INT, 4, ids, br), kc]
        check(ctx, cols, [0, 1], [], [1], [("count", 0), ("max", 0)])
        check(ctx, cols, [0, 1], [(0, GT, 0.0)], [1], [("count", 0), ("max", 0)])
        check(ctx, cols, [0, 1], [], [1], [("count", 0)])
        if key.dtype == np.int8:
            check(ctx, cols, [0, 1], [(0, LT, 500.0)], [1], [("max", 1), ("count", 0), ("min", 1)])
    # the same query handle run twice: the second run skips the form that overflowed
    seg = native.DeviceSegment(ctx, [RawColumn(DENSE_INT, 4, ids, br).native(), RawColumn(DENSE_STRING, 2, many, br).native()])
    q = native.DeviceQuery(ctx, seg, [0, 1], [], (), 0, 1024, group_cols=[1], aggs=[(native.AGG_COUNT, 0)])
    q.run()
    k1, f1, c1, _ = q.fetch_groups()
    q.run()
    k2, f2, c2, _ = q.fetch_groups()
    assert k1.size == 64 and (k1 == k2).all() and (f1 == f2).all() and (c1 == c2).all() and int(c1.sum()) == n
    q.close()
    seg.close()


# ---- cross-segment merge behind the C ABI: imm3_comm_merge_groups (ProjectAggregateQueueOp) ------------------------------------
def _decode_merged(ucols, group, aggs, keys, counts, vals):
    got = []
    for g in range(keys.shape[0]):
        raw = int(keys[g]).to_bytes(8, "little")
        parts, off = [], 0
        for gi in group:
            c = ucols[gi]
            chunk = raw[off: off + c.width]
            parts.append(chunk.decode() if c.codec == DENSE_STRING else str(int.from_bytes(chunk, "little", signed=True)))
            off += c.width
        st = []
        for j, (kind, ci) in enumerate(aggs):
            c = ucols[ci]
            if kind == "count":
                st.append(int(vals[g, j]))
            elif c.codec == DENSE_STRING:
                st.append(int(vals[g, j]).to_bytes(8, "big", signed=True)[8 - c.width:].decode())
            else:
                st.append(float(int(vals[g, j])))
        got.append(("_".join(parts), st))
    return got


@pytest.mark.parametrize("group,aggs", [
    ([2], [("count", 0), ("max", 1), ("min", 0)]),            # 2-byte key: direct table + element-wise all-reduces
    ([1], [("count", 1), ("max", 2)]),                        # 1-byte key, MAX over a string column (unsigned compare)
    ([], [("max", 0), ("count", 0)]),                         # no group column: one group
    ([1, 2], [("count", 0), ("min", 1), ("max", 0)]),         # 3-byte key: group lists exchanged with ncclAllGather, merged by key
    ([0], [("count", 0), ("max", 1)]),                        # int32 key
])
def test_merge_groups_across_segments_one_rank_rccl(ctx, group, aggs):
    """Three segments of different lengths on one rank, a real one-rank RCCL communicator: the merged table equals
    ProjectAggregateQueueOp's restatement (oracle_np.combine_agg) over the per-segment tables, in first-seen order."""
    rng = np.random.default_rng(31 + len(group))
    used = [0, 1, 2]
    sels = [(1, GT, -100.0)]
    segs, queries, per_seg = [], [], []
    for s, n in enumerate((70_000, 1, 33_333)):
        cols = make_cols(rng, n, blocks_of(n, 1024), id_range=40 if s != 1 else 3)
        _, _, masks = oracle_np.scan_select([c.npcol() for c in cols], sels, 1024)
        per_seg.append(oracle_np.project_agg([c.npcol() for c in cols], group, aggs, masks))
        seg = native.DeviceSegment(ctx, [c.native() for c in cols])
        q = native.DeviceQuery(ctx, seg, used, sels, (), 0, 1024, group_cols=group, aggs=[(KIND[k], c) for k, c in aggs])
        q.run()
        segs.append(seg)
        queries.append(q)
    expect = oracle_np.combine_agg(per_seg, aggs)
    comm = native.Comm(ctx, 1, 0, native.comm_unique_id())
    keys, first, counts, vals = comm.merge_groups(queries, [0, 1, 2])
    ucols = make_cols(np.random.default_rng(0), 1, [1])
    got = _decode_merged(ucols, group, aggs, keys, counts, vals)
    assert got == [(k, v) for k, v in expect.items()]
    assert (np.diff(first.astype(np.int64)) > 0).all() and int(counts.sum()) == sum(sum(v[[k for k, _ in aggs].index("count")] for v in seg.values()) for seg in per_seg)
    # the segment index decides the first-seen order: the same queries under reversed indices come out in another order
    keys2, first2, counts2, vals2 = comm.merge_groups(queries, [2, 1, 0])
    assert sorted(zip(keys.tolist(), counts.tolist())) == sorted(zip(keys2.tolist(), counts2.tolist()))
    with pytest.raises(native.Imm3Error):
        comm.merge_groups([], [])
    comm.close()
    (c0,) = native.Comm.create_all([ctx])                     # the single-process flavour (one JVM, every GPU): merged inside the process
    k3, f3, n3, v3 = native.Comm.merge_groups_all([c0], [queries], [[0, 1, 2]])
    assert k3.tolist() == keys.tolist() and f3.tolist() == first.tolist() and n3.tolist() == counts.tolist() and v3.tolist() == vals.tolist()
    c0.close()
    for q in queries:
        q.close()
    for s in segs:
        s.close()


def test_merge_of_a_hundred_thousand_groups_on_the_device(ctx):
    """Wide keys are merged on the device (a hash table: compare-and-swap on the key, atomics on the columns; csrc/imm3_agg.hip
    k_merge_*): three segments whose int32 group column takes ~130 000 distinct values, count + max(age) + min(id), against numpy --
    every group once, counts added, extremes combined, first arrival first (ascending segment index, then first row)."""
    rng = np.random.default_rng(77)
    used = [0, 1, 2]
    segs, queries = [], []
    all_ids, all_age, first_seen = [], [], {}
    for sidx, n in enumerate((260_000, 90_001, 300_123)):
        ids = rng.integers(0, 150_000, size=n).astype(np.int32)
        if sidx == 1:
            ids[:5] = -1                                              # (a key with every byte 0xFF: the hash table's empty marker once widened?  no: 4 bytes -- an ordinary key)
        age = rng.integers(-128, 128, size=n).astype(np.int8)
        st = np.array([list(CODES[i]) for i in rng.integers(0, len(CODES), size=n)], dtype=np.uint8).reshape(n, 2)
        br = blocks_of(n, 1024)
        cols = [RawColumn(DENSE_INT, 4, ids, br), RawColumn(DENSE_TINYINT, 1, age, br), RawColumn(DENSE_STRING, 2, st, br)]
        seg = native.DeviceSegment(ctx, [c.native() for c in cols])
        q = native.DeviceQuery(ctx, seg, used, [], (), 0, 1024, group_cols=[0], aggs=[(KIND["count"], 0), (KIND["max"], 1), (KIND["min"], 0)])
        q.run()
        segs.append(seg)
        queries.append(q)
        all_ids.append(ids)
        all_age.append(age)
        uniq, first = np.unique(ids, return_index=True)
        for k, f in zip(uniq.tolist(), first.tolist()):
            first_seen.setdefault(k, (sidx << 32) | f)
    comm = native.Comm(ctx, 1, 0, native.comm_unique_id())
    keys, first, counts, vals = comm.merge_groups(queries, [0, 1, 2])
    comm.close()
    ids = np.concatenate(all_ids)
    age = np.concatenate(all_age).astype(np.int64)
    uniq, inv, cnt = np.unique(ids, return_inverse=True, return_counts=True)
    mx = np.full(uniq.size, -(1 << 62), np.int64)
    np.maximum.at(mx, inv, age)
    got_keys = keys.astype(np.uint64).astype(np.uint32).view(np.int32) if keys.dtype != np.int32 else keys
    got_keys = (keys.astype(np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)
    assert keys.shape[0] == uniq.size > 100_000
    order = np.argsort(got_keys, kind="stable")
    assert (got_keys[order] == uniq).all()
    assert (counts[order].astype(np.int64) == cnt).all() and (vals[order, 0] == cnt).all()
    assert (vals[order, 1] == mx).all()
    assert (vals[order, 2] == uniq.astype(np.int64)).all()            # min(id) of the group id = the key
    assert (np.diff(first.astype(np.int64)) > 0).all()
    want_first = np.array([first_seen[k] for k in got_keys.tolist()], dtype=np.uint64)
    assert (first.astype(np.uint64) == want_first).all()
    for q in queries:
        q.close()
    for s in segs:
        s.close()
