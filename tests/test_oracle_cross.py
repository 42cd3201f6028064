"""The two independent CPU restatements (C faithful/tight and numpy) must agree bit-for-bit on
randomised tables -- layouts (uniform, ragged, loader-quirk, empty), predicates and projections."""
import numpy as np
import pytest

from conftest import DENSE_INT, DENSE_STRING, DENSE_TINYINT, EQ, GT, LT, MATCH, RawColumn, blocks_of
from oracle import oracle_c, oracle_np

CODES = [b"CA", b"NY", b"TX", b"WA", b"VA", b"DC", b"CT"]


def random_table(rng, n, block_rows):
    ids = rng.integers(-2**31, 2**31, size=n, dtype=np.int64).astype(np.int32)
    if rng.random() < 0.5:
        ids = rng.integers(-50, 50, size=n).astype(np.int32)
    age = rng.integers(-128, 128, size=n).astype(np.int8)
    st = np.array([list(CODES[i]) for i in rng.integers(0, len(CODES), size=n)], dtype=np.uint8).reshape(n, 2)
    return [RawColumn(DENSE_INT, 4, ids, block_rows), RawColumn(DENSE_TINYINT, 1, age, block_rows),
            RawColumn(DENSE_STRING, 2, st, block_rows)]


def random_sels(rng, cols):
    sels = []
    for _ in range(rng.integers(0, 5)):
        ci = int(rng.integers(0, len(cols)))
        c = cols[ci]
        if c.codec == DENSE_STRING:
            k = int(rng.integers(0, 4))
            vals = [CODES[i] for i in rng.integers(0, len(CODES), size=k)]
            if rng.random() < 0.3:
                vals.append(b"CAL")  # wrong length: can never match
            sels.append((ci, MATCH, vals))
        else:
            cond = [GT, LT, EQ][int(rng.integers(0, 3))]
            choice = rng.random()
            if choice < 0.6:
                v = float(rng.integers(-60, 60)) + float(rng.choice([0.0, 0.5, -0.5]))
            elif choice < 0.8:
                v = float(rng.choice([200.0, 128.0, 256.0, -129.0, 3e9, -3e9, 1e12, float("nan"), 2147483647.0, -2147483648.0]))
            else:
                v = float(rng.integers(-2**31, 2**31))
            sels.append((ci, cond, v))
    return sels


LAYOUTS = [
    (0, []), (1, [1]), (63, [63]), (64, [64]), (65, [65]), (100, [100]), (1024, [1024]), (1025, [1024, 1]),
    (2048 + 256, [1024, 1024, 256]), (300, [128, 128, 44]), (25, [4, 4, 1, 4, 4, 1, 4, 3]), (130, [64, 0, 66]),
    (200, [100, 100]), (5000, blocks_of(5000, 1024)), (777, blocks_of(777, 10)),
]


@pytest.mark.parametrize("n,block_rows", LAYOUTS)
def test_cross_layouts(n, block_rows):
    rng = np.random.default_rng(1234 + n)
    for trial in range(6):
        cols = random_table(rng, n, block_rows)
        order = rng.permutation(3)[: rng.integers(1, 4)]
        used = [cols[i] for i in order]
        sels = random_sels(rng, used)
        wc, cc = oracle_c.scan_select([c.ocol() for c in used], sels, 1024, flavour=trial % 2)
        wn, cn, masks = oracle_np.scan_select([c.npcol() for c in used], sels, 1024)
        assert cc == cn
        assert wc.tolist() == wn.tolist()
        size, oid, woff, tw = oracle_c.layout(used[0].ocol(), 1024)
        s2, o2, w2, tw2 = oracle_np.layout(used[0].offsets, used[0].width, 1024)
        assert size.tolist() == s2.tolist() == list(block_rows) and oid.tolist() == o2.tolist()
        assert woff.tolist() == w2.tolist() and tw == tw2
        proj = [int(i) for i in rng.permutation(len(used))[: rng.integers(1, len(used) + 1)]]
        limit = int(rng.choice([0, 0, 1, 7, 10, 10**6]))
        n_out, batch, pos, vals, wt = oracle_c.project([c.ocol() for c in used], proj, limit, 1024, wc)
        rows, where, wt2 = oracle_np.project([c.npcol() for c in used], proj, limit, masks)
        assert n_out == len(rows)
        assert list(zip(batch.tolist(), pos.tolist())) == where
        if limit == 0:
            assert wt == wt2
        for j, pj in enumerate(proj):
            c = used[pj]
            if c.codec == DENSE_INT:
                got = vals[j].view("<i4").reshape(-1).tolist()
            elif c.codec == DENSE_TINYINT:
                got = vals[j].view(np.int8).reshape(-1).tolist()
            else:
                got = [bytes(r) for r in vals[j]]
            assert got == [r[j] for r in rows]


def test_faithful_equals_tight_large():
    rng = np.random.default_rng(7)
    n = 200_000
    cols = random_table(rng, n, blocks_of(n, 1024))
    sels = [(1, GT, 18.0), (1, LT, 30.0), (0, GT, -1e9), (2, MATCH, [b"CA", b"NY"])]
    w0, c0 = oracle_c.scan_select([c.ocol() for c in cols], sels, 1024, 0)
    w1, c1 = oracle_c.scan_select([c.ocol() for c in cols], sels, 1024, 1)
    assert c0 == c1 and (w0 == w1).all()
    wn, cn, _ = oracle_np.scan_select([c.npcol() for c in cols], sels, 1024)
    assert cn == c0 and (wn == w0).all()


def test_partial_trailing_element_stale_bytes():
    """bytes % width != 0: read() returns a short count that the codec ignores, so the last value re-uses
    the previous chunk's tail (DenseCodec.scala:41; SURVEY A.1 rule 2).  Both restatements agree."""
    raw = np.frombuffer(bytes([1, 0, 0, 0, 2, 0, 0, 0x7F, 9]), dtype=np.uint8)
    c = oracle_c.OColumn(raw, np.array([0, 9], np.int32), DENSE_INT, 4)
    words, count = oracle_c.scan_select([c], [(0, GT, 5.0)], 1024)
    v = oracle_np.decode_block(raw, DENSE_INT, 4)
    assert v.tolist() == [1, 0x7F000002, 0x7F000009]
    wn, cn, _ = oracle_np.scan_select([(raw, np.array([0, 9], np.int32), DENSE_INT, 4)], [(0, GT, 5.0)], 1024)
    assert count == cn == 2 and words.tolist() == wn.tolist() == [0b110]


def test_zero_batches_errors():
    """NotMatch/NoOp throw when the chain is built; wrong vector type only when a batch is processed."""
    empty = oracle_c.OColumn(np.zeros(0, np.uint8), np.array([0], np.int32), DENSE_STRING, 2)
    with pytest.raises(oracle_c.OracleError):
        oracle_c.scan_select([empty], [(0, 1, [b"CA"])], 1024)
    words, count = oracle_c.scan_select([empty], [(0, GT, 1.0)], 1024)   # no batch -> never evaluated
    assert count == 0 and words.size == 0
    with pytest.raises(oracle_np.RefException):
        oracle_np.scan_select([(np.zeros(0, np.uint8), np.array([0], np.int32), DENSE_STRING, 2)], [(0, 1, [b"CA"])], 1024)
    wn, cn, _ = oracle_np.scan_select([(np.zeros(0, np.uint8), np.array([0], np.int32), DENSE_STRING, 2)], [(0, GT, 1.0)], 1024)
    assert cn == 0


def test_project_agg_c_twin_agrees_with_numpy(oracle):
    """Group-by aggregation is restated twice (C: imm3o_project_agg, numpy: oracle_np.project_agg): same groups, same
    first-seen order, same counts / extremes on random tables, layouts, group columns and aggregate lists."""
    from oracle import oracle_np
    rng = np.random.default_rng(77)
    codes = [b"CA", b"NY", b"TX", b"WA", b"VA", b"DC", b"CT", b"a_", b"_b"]
    for trial in range(30):
        n = int(rng.integers(1, 700))
        block = int(rng.choice([64, 100, 128, 1024]))
        br = blocks_of(n, block)
        ids = rng.integers(-40, 40, size=n).astype(np.int32)
        age = rng.integers(-128, 128, size=n).astype(np.int8)
        st = np.array([list(codes[i]) for i in rng.integers(0, len(codes), size=n)], dtype=np.uint8).reshape(n, 2)
        cols = [RawColumn(DENSE_INT, 4, ids, br), RawColumn(DENSE_TINYINT, 1, age, br), RawColumn(DENSE_STRING, 2, st, br)]
        sels = [] if trial % 3 == 0 else [(0, GT, float(rng.integers(-40, 20)))]
        words, _ = oracle.scan_select([c.ocol() for c in cols], sels, block)
        _, _, masks = oracle_np.scan_select([c.npcol() for c in cols], sels, block)
        group = [int(g) for g in rng.permutation(3)[: int(rng.integers(0, 3))]]
        pool = [("count", 0), ("count", 2), ("max", 0), ("min", 0), ("max", 1), ("min", 1), ("max", 2)]
        aggs = [pool[i] for i in rng.permutation(len(pool))[: int(rng.integers(1, 5))]]
        a = oracle.project_agg([c.ocol() for c in cols], group, aggs, words)
        b = oracle_np.project_agg([c.npcol() for c in cols], group, aggs, masks)
        assert list(a.items()) == list(b.items()), (trial, group, aggs)
    # a String vector only takes CountAggr / MaxStringAggr
    with pytest.raises(oracle.OracleError, match="bad aggregator"):
        oracle.project_agg([c.ocol() for c in cols], [0], [("min", 2)], oracle.scan_select([c.ocol() for c in cols], [], block)[0])
