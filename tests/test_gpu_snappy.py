"""Snappy-coded columns on the GPU (csrc/imm3_snappy.hip) against the oracle: the GPU reads blocks in the format
SnappyCodec.encode writes (core/codec/SnappyCodec.scala:15-43), the oracle runs the same query over the DENSE_* column
holding the same values in the same blocks.  Blocks come from two independent compressors (the oracle's and pyarrow's
Google snappy inside the same framing); bit-exact bitmap, count, row order and projected values; corrupt blocks and
checksum mismatches are errors."""
import os

import numpy as np
import pytest

from conftest import (DENSE_INT, DENSE_STRING, DENSE_TINYINT, EQ, GT, LT, MATCH, PforColumn, RawColumn, SnappyColumn, blocks_of)
from test_gpu_parity import check, ctx  # noqa: F401  (ctx is a fixture)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODES = [b"CA", b"NY", b"TX", b"WA", b"VA", b"DC", b"CT"]


def columns(rng, n, block_rows, encoder):
    br = blocks_of(n, block_rows)
    ids = SnappyColumn(DENSE_INT, 4, (np.arange(n, dtype=np.int64) // 3 * 7 - 1000).astype(np.int32), br, encoder)       # runs: compressible
    noise = SnappyColumn(DENSE_INT, 4, rng.integers(-2**31, 2**31, n).astype(np.int32), br, encoder)                    # stored chunks
    age = SnappyColumn(DENSE_TINYINT, 1, rng.integers(0, 100, n).astype(np.int8), br, encoder)
    st = SnappyColumn(DENSE_STRING, 2, np.array([list(CODES[i]) for i in rng.integers(0, len(CODES), n)], dtype=np.uint8).reshape(n, 2), br, encoder)
    wide = SnappyColumn(DENSE_STRING, 5, np.array([list(b"ab%03d" % (i % 17)) for i in range(n)], dtype=np.uint8).reshape(n, 5), br, encoder)
    return [ids, noise, age, st, wide]


@pytest.mark.parametrize("encoder", ["oracle", "google"])
@pytest.mark.parametrize("n,block_rows", [(5000, 1024), (1024 * 9 + 1, 1024), (777, 100), (40000, 7000), (3, 1024), (70000, 1024)])
def test_snappy_columns_match_dense(ctx, oracle, n, block_rows, encoder):
    rng = np.random.default_rng(n)
    cols = columns(rng, n, block_rows, encoder)
    q = float(np.quantile(cols[0].values.astype(np.float64), 0.4))
    check(ctx, oracle, cols, [0], [(0, GT, q)], proj=[0], block_size=block_rows)
    check(ctx, oracle, cols, [2, 0, 3], [(0, GT, 18.0), (0, LT, 60.0), (1, GT, q), (2, MATCH, [b"CA", b"NY"])], proj=[1, 2, 0], block_size=block_rows)
    check(ctx, oracle, cols, [1, 4], [(0, LT, 0.0), (1, MATCH, [b"ab003", b"ab016"])], proj=[1, 0], limit=50, block_size=block_rows)
    check(ctx, oracle, cols, [3], [(0, MATCH, [b"TX"])], block_size=block_rows)


def test_mixed_codecs_in_one_segment(ctx, oracle):
    rng = np.random.default_rng(3)
    n = 1024 * 6 + 10
    br = blocks_of(n, 1024)
    cols = [PforColumn(np.arange(n, dtype=np.int32) * 2, br), SnappyColumn(DENSE_TINYINT, 1, rng.integers(0, 100, n).astype(np.int8), br),
            RawColumn(DENSE_STRING, 2, np.array([list(CODES[i]) for i in rng.integers(0, 7, n)], dtype=np.uint8), br)]
    check(ctx, oracle, cols, [0, 1, 2], [(0, GT, 500.0), (1, LT, 50.0), (2, MATCH, [b"CA"])], proj=[2, 1, 0])
    check(ctx, oracle, cols, [1, 0], [(0, GT, 90.0), (1, LT, 9000.0)])


def test_aggregate_over_snappy(ctx, oracle):
    from test_gpu_agg import check as agg_check
    rng = np.random.default_rng(9)
    n = 1024 * 5 + 77
    br = blocks_of(n, 1024)
    st = SnappyColumn(DENSE_STRING, 2, np.array([list(CODES[i]) for i in rng.integers(0, 7, n)], dtype=np.uint8), br)
    val = SnappyColumn(DENSE_INT, 4, rng.integers(-10**6, 10**6, n).astype(np.int32), br)
    agg_check(ctx, [st, val], [0, 1], [(1, GT, 0.0)], [0], [("count", 0), ("min", 1), ("max", 1), ("max", 0)])


def test_corrupt_blocks_are_errors(ctx, oracle):
    from immutable3_amd import native
    n = 3000
    col = SnappyColumn(DENSE_INT, 4, np.arange(n, dtype=np.int32) // 5, blocks_of(n, 1024))
    o1 = int(col.offsets[1])

    def query(dat):
        seg = native.DeviceSegment(ctx, [(col.codec, 4, dat, dat.size, col.offsets)])
        try:
            q = native.DeviceQuery(ctx, seg, [0], [(0, GT, 5.0)])
            q.run()
            return q.count()
        finally:
            seg.close()

    assert query(col.dat.copy()) == int((col.values > 5).sum())
    for pos, what in ((o1 + 0, "stream header"), (o1 + 7, "flag"), (o1 + 11, "checksum"), (o1 + 20, "payload byte")):
        dat = col.dat.copy()
        dat[pos] ^= 0x40
        with pytest.raises(native.Imm3Error):
            query(dat)


def test_engines_over_snappy_table(tmp_path):
    """Python Engine and the C++ CLI over a table whose three columns are snappy-coded vs the same table stored DENSE."""
    import subprocess
    from immutable3_amd import GT, LT, And, Match, Project, Query, Select, synth
    from immutable3_amd.operators import Engine, GpuSegmentManager
    from immutable3_amd.schema import CodecType, Column, Table, TableIO
    from immutable3_amd.storage import SegmentManager, write_segment_arrays
    for name, codecs in (("ts", (CodecType.SNAPPY_INT, CodecType.SNAPPY_STRING, CodecType.SNAPPY_TINYINT)),
                         ("td", (CodecType.DENSE_INT, CodecType.DENSE_STRING, CodecType.DENSE_TINYINT))):
        t = Table(name, [Column.make("id", codecs[0]), Column.make("state", codecs[1], {"size": "2"}), Column.make("age", codecs[2])], 1024)
        TableIO.store(str(tmp_path), t)
        for s in range(3):
            n = 5000 + 300 * s
            write_segment_arrays(str(tmp_path), t, s, {"id": (np.arange(n, dtype=np.int64) // 2 + s * 10 ** 6).astype(np.int32),
                                                       "age": synth.uniform_below(70 + s, n, 100, np.int8), "state": synth.state_codes(80 + s, n)})
    g = GpuSegmentManager(SegmentManager(str(tmp_path)))
    e = Engine(g)
    sel = And(And(Select("id", GT(1000)), Select("id", LT(1_002_000))), And(Select("age", LT(30)), Select("state", Match(["CA", "NY"]))))
    for proj in (Project(["id", "state", "age"]), Project(["age", "id"], 40)):
        a = [tuple(r) for r in e.execute(Query("ts", sel, proj))]
        b = [tuple(r) for r in e.execute(Query("td", sel, proj))]
        assert a == b and len(a) > 0
    g.close()
    exe = os.path.join(ROOT, "immutable3_amd", "bin", "imm3_sql")
    for sql in ("select id, state, age from {t} where (id > 1000 and id < 1002000 and age < 30 and state = 'CA')",
                "select state, id from {t} limit 5",
                "select count(id), max(age), max(state) from {t} where age > 10 group by state"):
        outs = []
        for tn in ("ts", "td"):
            p = subprocess.run([exe, "-q", sql.format(t=tn), "-d", str(tmp_path)], capture_output=True, text=True)
            assert p.returncode == 0, p.stdout + p.stderr
            outs.append(p.stdout.splitlines())
        assert outs[0] == outs[1] and len(outs[0]) > 0, sql


def test_random_corruption_never_faults(ctx, oracle):
    """Flipped bits anywhere in a compressed column: the query either fails with an error or returns normally (PFOR_INT
    carries no checksum, so a flipped payload bit may just change values) -- it never reads or writes out of bounds."""
    from immutable3_amd import native
    rng = np.random.default_rng(123)
    n = 1024 * 3 + 200
    br = blocks_of(n, 1024)
    base = [SnappyColumn(DENSE_INT, 4, (np.arange(n) // 3).astype(np.int32), br),
            SnappyColumn(DENSE_STRING, 2, np.array([list(CODES[i % 7]) for i in range(n)], dtype=np.uint8), br, "google"),
            PforColumn((np.arange(n) * 5).astype(np.int32), br),
            PforColumn(rng.integers(-2**31, 2**31, n).astype(np.int32), br)]
    errors = 0
    for it in range(24):
        col = base[it % len(base)]
        dat = col.dat.copy()
        for _ in range(1 + it % 3):
            dat[int(rng.integers(0, dat.size))] ^= np.uint8(1 << int(rng.integers(0, 8)))
        try:
            seg = native.DeviceSegment(ctx, [(col.codec, col.width, dat, dat.size, col.offsets)])
        except native.Imm3Error:
            errors += 1
            continue
        try:
            sel = [(0, MATCH, [b"CA"])] if col.width == 2 else [(0, GT, 100.0)]
            for proj in ([], [0]):
                q = native.DeviceQuery(ctx, seg, [0], sel, proj)
                q.run()
                q.count()
                q.close()
        except native.Imm3Error:
            errors += 1
        finally:
            seg.close()
    assert errors >= 6   # every flipped snappy bit is caught by the framing checks or the CRC
