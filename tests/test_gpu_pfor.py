"""PFOR_INT columns on the GPU (csrc/imm3_codec.hip) against the oracle: the GPU reads the blocks the reference's
ENCODER writes (oracle restatement of PFORCodec.scala:19-31), the oracle runs the same query over the DENSE_INT column
holding the same values in the same blocks.  Bit-exact bitmap, count, row order and projected values.
Two device paths are covered: k_filter_pfor (predicate on the compressed blocks; tile-aligned segments, the column not
projected) and k_pfor_decode (decoded column; projection, aggregation, ragged or odd block sizes, table queries)."""
import os

import numpy as np
import pytest

from conftest import DENSE_INT, DENSE_STRING, DENSE_TINYINT, EQ, GT, LT, MATCH, PforColumn, RawColumn, blocks_of
from test_gpu_parity import check, ctx  # noqa: F401  (ctx is a fixture)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def value_cases(rng, n):
    return {
        "sorted_ids": np.arange(n, dtype=np.int64) * 3 - 1000,
        "random": rng.integers(-2**31, 2**31, n),
        "small_deltas": np.cumsum(rng.integers(0, 4, n)),
        "constant": np.full(n, 123456),
        "medium_deltas": np.cumsum(rng.integers(0, 2**18, n)) - 2**30,
        "mixed_raw_packed": np.where(rng.random(n) < 0.01, -1, 1).cumsum() * 7,
        "wide_sorted": np.sort(rng.integers(-2**31, 2**31, n)),
    }


@pytest.mark.parametrize("n", [1024, 5000, 1024 * 37 + 256, 1024 * 8 + 1, 1024 * 3 + 31, 1024 * 2 + 33, 1000, 17])
def test_fused_filter_tile_aligned(ctx, oracle, n):
    rng = np.random.default_rng(n)
    for name, v in value_cases(rng, n).items():
        v = v.astype(np.int64).astype(np.int32)
        col = PforColumn(v, blocks_of(n, 1024))
        lo, hi = np.quantile(v.astype(np.float64), [0.3, 0.7])
        for sels in ([(0, GT, float(lo)), (0, LT, float(hi))], [(0, EQ, float(v[n // 2]))], [(0, LT, float(lo))], [(0, GT, 3e9)]):
            check(ctx, oracle, [col], [0], sels)


def test_fused_filter_with_other_columns(ctx, oracle):
    rng = np.random.default_rng(5)
    n = 1024 * 20 + 100
    ids = PforColumn(np.arange(n, dtype=np.int32) * 2, blocks_of(n, 1024))
    age = RawColumn(DENSE_TINYINT, 1, rng.integers(0, 100, n).astype(np.int8), blocks_of(n, 1024))
    st = RawColumn(DENSE_STRING, 2, np.array([list(c) for c in rng.choice([b"CA", b"NY", b"TX"], n)], dtype=np.uint8), blocks_of(n, 1024))
    v2 = PforColumn(rng.integers(0, 1000, n).astype(np.int32), blocks_of(n, 1024))
    cols = [ids, age, st, v2]
    # PFOR predicate + int8 predicate + string predicate, projecting the non-PFOR columns (PFOR stays fused)
    check(ctx, oracle, cols, [0, 1, 2], [(0, GT, 1000.0), (0, LT, 30000.0), (1, GT, 18.0), (1, LT, 30.0), (2, MATCH, [b"CA"])], proj=[1, 2])
    # two PFOR predicates (two fused passes ANDed) with the int8 column first
    check(ctx, oracle, cols, [1, 0, 3], [(0, GT, 50.0), (1, GT, 5000.0), (2, LT, 500.0)])
    # the PFOR column is projected: decoded path, predicate evaluated on the decoded column
    check(ctx, oracle, cols, [0, 1], [(0, GT, 1000.0), (0, LT, 30000.0), (1, LT, 10.0)], proj=[0, 1])
    # no predicate at all, project the PFOR columns with a limit
    check(ctx, oracle, cols, [0, 3], [], proj=[1, 0], limit=777)


@pytest.mark.parametrize("block_rows", [1000, 100, 64, 33, 2048, 4096 + 77, 3000])
def test_decoded_path_odd_blocks(ctx, oracle, block_rows):
    rng = np.random.default_rng(block_rows)
    n = block_rows * 5 + max(1, block_rows // 3)
    for name, v in value_cases(rng, n).items():
        v = v.astype(np.int64).astype(np.int32)
        col = PforColumn(v, blocks_of(n, block_rows))
        lo, hi = np.quantile(v.astype(np.float64), [0.25, 0.75])
        check(ctx, oracle, [col], [0], [(0, GT, float(lo)), (0, LT, float(hi))], proj=[0], block_size=block_rows)


def test_loader_quirk_trailing_one_row_block(ctx, oracle):
    # a "full" segment ends with a 1-row block -> a PFOR block that is variable-byte only
    v = (np.arange(2049, dtype=np.int32) * 5) - 7
    col = PforColumn(v, [1024, 1024, 1])
    check(ctx, oracle, [col], [0], [(0, GT, 100.0)], proj=[0])
    check(ctx, oracle, [col], [0], [(0, GT, 100.0)])


def test_aggregate_over_pfor(ctx, oracle):
    from test_gpu_agg import check as agg_check
    rng = np.random.default_rng(9)
    n = 1024 * 12 + 5
    kcol = RawColumn(DENSE_TINYINT, 1, rng.integers(0, 7, n).astype(np.int8), blocks_of(n, 1024))
    vcol = PforColumn(rng.integers(-10**6, 10**6, n).astype(np.int32), blocks_of(n, 1024))
    gcol = PforColumn(np.sort(rng.integers(0, 20, n)).astype(np.int32), blocks_of(n, 1024))
    # aggregate over a PFOR column, predicate on it too (the predicate then reads the decoded column)
    agg_check(ctx, [kcol, vcol], [0, 1], [(1, GT, 0.0)], [0], [("count", 0), ("min", 1), ("max", 1)])
    # group by a PFOR column; predicate on another PFOR column stays on its compressed blocks
    agg_check(ctx, [gcol, vcol], [0, 1], [(1, LT, 5000.0)], [0], [("count", 0)])


def test_table_query_over_pfor_segments(ctx, oracle):
    from immutable3_amd import native
    rng = np.random.default_rng(11)
    segs, ocounts = [], 0
    for s in range(3):
        n = 1024 * 4 + 100 * s
        v = (np.arange(n, dtype=np.int32) + s * 10**6)
        col = PforColumn(v, blocks_of(n, 1024))
        segs.append(native.DeviceSegment(ctx, [col.native()]))
        ocounts += int(((v > 1000) & (v < 2 * 10**6 + 50)).sum())
    t = native.DeviceTable(ctx, segs)
    q = native.DeviceQuery(ctx, t, [0], [(0, GT, 1000.0), (0, LT, 2.0 * 10**6 + 50)])
    q.run()
    assert q.count() == ocounts
    q.close()
    t.close()
    for s in segs:
        s.close()


def test_malformed_block_is_an_error(ctx, oracle):
    from immutable3_amd import native
    v = np.arange(2048, dtype=np.int32)
    col = PforColumn(v, [1024, 1024])
    dat = col.dat.copy()
    o1 = int(col.offsets[1])
    dat[o1 + 4] = 40  # first width of the second block's first group header (big-endian high byte) -> 40 > 32
    seg = native.DeviceSegment(ctx, [(col.codec, 4, dat, dat.size, col.offsets)])
    q = native.DeviceQuery(ctx, seg, [0], [(0, GT, 5.0)])
    q.run()
    with pytest.raises(native.Imm3Error):
        q.count()
    q.close()
    with pytest.raises(native.Imm3Error):  # decoded path reports at decode time
        native.DeviceQuery(ctx, seg, [0], [(0, GT, 5.0)], [0])
    seg.close()
    # count word disagreeing with the layout
    dat = col.dat.copy()
    dat[3] = 7
    seg = native.DeviceSegment(ctx, [(col.codec, 4, dat, dat.size, col.offsets)])
    with pytest.raises(native.Imm3Error):
        q = native.DeviceQuery(ctx, seg, [0], [(0, GT, 5.0)], [0])
    seg.close()


def _two_tables(tmp_path, n_segs=3):
    """The same rows twice: table `tp` stores id as PFOR_INT, table `td` as DENSE_INT."""
    from immutable3_amd import synth
    from immutable3_amd.schema import CodecType, Column, Table, TableIO
    from immutable3_amd.storage import write_segment_arrays
    tabs = {}
    for name, codec in (("tp", CodecType.PFOR_INT), ("td", CodecType.DENSE_INT)):
        t = Table(name, [Column.make("id", codec), Column.make("state", CodecType.DENSE_STRING, {"size": "2"}),
                         Column.make("age", CodecType.DENSE_TINYINT)], 1024)
        TableIO.store(str(tmp_path), t)
        for s in range(n_segs):
            n = 5000 + 300 * s
            cols = {"id": (np.arange(n, dtype=np.int64) * 3 + s * 10 ** 6).astype(np.int32),
                    "age": synth.uniform_below(70 + s, n, 100, np.int8), "state": synth.state_codes(80 + s, n)}
            write_segment_arrays(str(tmp_path), t, s, cols)
        tabs[name] = t
    return tabs


def test_python_engine_over_pfor_table(tmp_path):
    from immutable3_amd import GT, LT, And, Match, Project, Query, Select
    from immutable3_amd.operators import Engine, GpuSegmentManager, ProjectOp, ScanOp, SelectOp
    from immutable3_amd.storage import SegmentManager
    _two_tables(tmp_path)
    g = GpuSegmentManager(SegmentManager(str(tmp_path)))
    e = Engine(g)
    sel = And(And(Select("id", GT(2000)), Select("id", LT(1_009_000))), Select("age", LT(30)))
    for proj in (Project(["id", "age"]), Project(["age", "state"], 50), Project(["id"], 7)):
        rows = {}
        for tn in ("tp", "td"):
            rows[tn] = [tuple(r) for r in e.execute(Query(tn, sel, proj))]
        assert rows["tp"] == rows["td"] and len(rows["tp"]) > 0
    # operator level: the batches' vectors of a PFOR_INT column are the GPU's decode
    for seg in range(2):
        ba = list(SelectOp("id", GT(6000), ScanOp(g, seg, "tp", [g.sm.getTable("tp").getColumn("id")])).iterator())
        bb = list(SelectOp("id", GT(6000), ScanOp(g, seg, "td", [g.sm.getTable("td").getColumn("id")])).iterator())
        assert len(ba) == len(bb)
        for x, y in zip(ba, bb):
            assert x.size == y.size and x.selected.words.tolist() == y.selected.words.tolist()
            assert np.asarray(x.columnVectors[0].data).tolist() == np.asarray(y.columnVectors[0].data).tolist()
    g.close()


def test_sql_cli_over_pfor_table(tmp_path):
    import subprocess
    _two_tables(tmp_path)
    exe = os.path.join(ROOT, "immutable3_amd", "bin", "imm3_sql")

    def run(sql):
        p = subprocess.run([exe, "-q", sql, "-d", str(tmp_path)], capture_output=True, text=True)
        assert p.returncode == 0, p.stdout + p.stderr
        return p.stdout.splitlines()

    for sql in ("select id, age from {t} where (id > 2000 and id < 1009000 and age < 30)",
                "select state, id from {t} where (id > 14000 and state = 'CA') limit 25",
                "select id from {t} limit 3",
                "select count(id), max(id), min(age) from {t} where id > 9000 group by state",
                "select count(age) from {t} where age > 50 group by id"):
        a, b = run(sql.format(t="tp")), run(sql.format(t="td"))
        assert a == b and len(a) > 0, sql
