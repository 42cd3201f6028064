"""PFOR_INT columns on the GPU (csrc/imm3_codec.hip) against the oracle: the GPU reads the blocks the reference's
ENCODER writes (oracle restatement of PFORCodec.scala:19-31), the oracle runs the same query over the DENSE_INT column
holding the same values in the same blocks.  Bit-exact bitmap, count, row order and projected values.
Two device paths are covered: k_filter_pfor (predicate on the compressed blocks; tile-aligned segments, the column not
projected) and k_pfor_decode (decoded column; projection, aggregation, ragged or odd block sizes, table queries)."""
import numpy as np
import pytest

from conftest import DENSE_INT, DENSE_STRING, DENSE_TINYINT, EQ, GT, LT, MATCH, PforColumn, RawColumn, blocks_of
from test_gpu_parity import check, ctx  # noqa: F401  (ctx is a fixture)

pytestmark = pytest.mark.gpu


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
