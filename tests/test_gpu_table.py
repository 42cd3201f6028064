"""GPU suite: table-level queries (imm3_table): ONE launch over the tile table of all segments must give exactly
the per-segment results concatenated in segment order -- bitmaps per segment, global count, rows in (segment, row)
order with a global limit, groups merged in first-seen order."""
import numpy as np
import pytest

from conftest import DENSE_INT, DENSE_STRING, DENSE_TINYINT, EQ, GT, LT, MATCH, RawColumn, blocks_of
from immutable3_amd import native
from oracle import oracle_np

pytestmark = pytest.mark.gpu
CODES = [b"CA", b"NY", b"TX", b"WA", b"VA", b"DC", b"CT"]
KIND = {"count": native.AGG_COUNT, "min": native.AGG_MIN, "max": native.AGG_MAX}


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0)
    yield c
    c.close()


def make_segment(rng, n, block_rows):
    ids = rng.integers(-50, 50, size=n).astype(np.int32)
    age = rng.integers(-128, 128, size=n).astype(np.int8)
    st = np.array([list(CODES[i]) for i in rng.integers(0, len(CODES), size=n)], dtype=np.uint8).reshape(n, 2)
    return [RawColumn(DENSE_INT, 4, ids, block_rows), RawColumn(DENSE_TINYINT, 1, age, block_rows), RawColumn(DENSE_STRING, 2, st, block_rows)]


# segment shapes: loader-quirk segments (S*B + 1 rows, trailing 1-row block), partial last blocks, a tiny and an empty one
SHAPES = [
    [(8 * 1024 + 1, [1024] * 8 + [1]), (8 * 1024 + 1, [1024] * 8 + [1]), (3 * 1024 + 700, [1024] * 3 + [700])],
    [(100, [100]), (0, []), (1, [1]), (64, [64]), (5000, blocks_of(5000, 1024))],
    [(70000, blocks_of(70000, 1024)), (1024, [1024]), (2048, [1024, 1024]), (1025, [1024, 1])],
    [(4 * 64 + 5, [64, 128, 64, 5])],
]


@pytest.mark.parametrize("shape", SHAPES)
def test_table_query_equals_per_segment(ctx, oracle, shape):
    rng = np.random.default_rng(len(shape) * 31 + shape[0][0])
    segs_cols = [make_segment(rng, n, br) for n, br in shape]
    dsegs = [native.DeviceSegment(ctx, [c.native() for c in cols]) for cols in segs_cols]
    table = native.DeviceTable(ctx, dsegs)
    queries = [
        ([1, 0], [(0, GT, -20.0), (0, LT, 60.0), (1, GT, -10.0)], [1, 0], 0),
        ([2, 0, 1], [(0, MATCH, [b"CA", b"TX"])], [1, 0, 2], 0),
        ([0], [], [0], 7),
        ([1, 2], [(0, EQ, 5.0), (1, MATCH, [b"NY"])], [0], 0),
        ([0, 1, 2], [(0, GT, 0.0), (1, LT, 0.0), (2, MATCH, [b"CA", b"NY", b"TX", b"WA"])], [2, 1], 1000),
    ]
    for used, sels, proj, limit in queries:
        q = native.DeviceQuery(ctx, table, used, sels, proj, limit, 1024)
        q.run()
        words, count = q.bitmap(), q.count()
        size, oid, woff = q.batches()
        fb, fw = q.segment_starts()
        idx, vals = q.fetch_rows()
        seg_of, row_of = q.locate_rows(idx)
        q.close()
        exp_rows = []
        total = 0
        for si, cols in enumerate(segs_cols):
            ucols = [cols[i] for i in used]
            ow, oc = oracle.scan_select([c.ocol() for c in ucols], sels, 1024, 1)
            osize, ooid, owoff, _ = oracle.layout(ucols[0].ocol(), 1024)
            total += oc
            b0, b1 = int(fb[si]), int(fb[si + 1])
            assert size[b0:b1].tolist() == osize.tolist() and oid[b0:b1].tolist() == ooid.tolist()
            assert (woff[b0:b1] - fw[si]).tolist() == owoff.tolist()
            assert words[int(fw[si]): int(fw[si]) + ow.size].tolist() == ow.tolist()
            assert not words[int(fw[si]) + ow.size: int(fw[si + 1])].any()          # padding up to the next tile
            n, batch, pos, ovals, _ = oracle.project([c.ocol() for c in ucols], proj, 0, 1024, ow)
            starts = np.concatenate([[0], np.cumsum(osize.astype(np.int64))])
            for r in range(n):
                exp_rows.append((si, int(starts[batch[r]] + pos[r]), [bytes(v[r]) for v in ovals]))
        assert count == total
        if limit > 0:
            exp_rows = exp_rows[:limit]
        assert idx.shape[0] == len(exp_rows)
        assert seg_of.tolist() == [e[0] for e in exp_rows] and row_of.tolist() == [e[1] for e in exp_rows]
        for j in range(len(proj)):
            assert [bytes(v) for v in vals[j]] == [e[2][j] for e in exp_rows]
    # aggregation over the table == per-segment aggregation merged in segment order
    for used, sels, group, aggs in [([2, 1, 0], [], [0], [("count", 2), ("max", 1), ("min", 2)]),
                                    ([1, 0], [(0, GT, 0.0)], [], [("count", 0), ("max", 1)]),
                                    ([0, 2], [(1, MATCH, [b"CA", b"DC"])], [0, 1], [("count", 0)])]:
        q = native.DeviceQuery(ctx, table, used, sels, (), 0, 1024, group_cols=group, aggs=[(KIND[k], c) for k, c in aggs])
        q.run()
        keys, first, counts, vals = q.fetch_groups()
        q.close()
        per_seg = []
        for cols in segs_cols:
            ucols = [cols[i] for i in used]
            _, _, masks = oracle_np.scan_select([c.npcol() for c in ucols], sels, 1024)
            per_seg.append(oracle_np.project_agg([c.npcol() for c in ucols], group, aggs, masks))
        expect = oracle_np.combine_agg(per_seg, aggs)
        got = []
        ucols = [segs_cols[0][i] for i in used]
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
                st.append(int(counts[g]) if kind == "count" else float(int(vals[g, j])))
            got.append(("_".join(parts), st))
        assert got == [(k, v) for k, v in expect.items()]
    table.close()
    for d in dsegs:
        d.close()


def test_table_rejects_ragged_and_mismatched(ctx):
    rng = np.random.default_rng(3)
    a = make_segment(rng, 25, [4, 4, 1, 4, 4, 1, 4, 3])            # non-final blocks not multiples of 64
    b = make_segment(rng, 100, [100])
    da, db = native.DeviceSegment(ctx, [c.native() for c in a]), native.DeviceSegment(ctx, [c.native() for c in b])
    with pytest.raises(native.Imm3Error) as e:
        native.DeviceTable(ctx, [db, da])
    assert e.value.code == native.ERR_LAYOUT
    dc = native.DeviceSegment(ctx, [b[0].native(), b[1].native()])  # different column set
    with pytest.raises(native.Imm3Error) as e:
        native.DeviceTable(ctx, [db, dc])
    assert e.value.code == native.ERR_ARG
    t = native.DeviceTable(ctx, [db])
    with pytest.raises(native.Imm3Error):                           # width-2 strings only on the table path
        wide = [RawColumn(DENSE_STRING, 3, np.zeros((10, 3), np.uint8), [10])]
        dw = native.DeviceSegment(ctx, [c.native() for c in wide])
        tw = native.DeviceTable(ctx, [dw])
        q = native.DeviceQuery(ctx, tw, [0], [(0, MATCH, [b"abc"])])
        q.run()
    t.close(); da.close(); db.close(); dc.close()
