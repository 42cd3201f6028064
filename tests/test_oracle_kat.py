"""Pins BOTH CPU restatements to the hand-derived known-answer vectors of SURVEY.md Appendix B.
The reference ships no tests, fixtures or golden vectors: these KATs, each derived
from a cited reference line, are the pin."""
import numpy as np
import pytest

from conftest import DENSE_INT, DENSE_STRING, DENSE_TINYINT, EQ, GT, LT, MATCH, NOOP, NOTMATCH, RawColumn, blocks_of
from oracle import oracle_c, oracle_np


# B1 int32 LE  (DataType.scala:40-47, Conversions.scala:17-24)
B1 = [(0, "00000000"), (1, "01000000"), (258, "02010000"), (-1, "ffffffff"), (-2147483648, "00000080"),
      (2147483647, "ffffff7f"), (0x12345678, "78563412"), (-123456789, "eb32a4f8")]


@pytest.mark.parametrize("value,hexbytes", B1)
def test_b1_int32_le(value, hexbytes):
    b = bytes.fromhex(hexbytes)
    assert oracle_c.int_to_bytes(value) == b
    assert oracle_c.bytes_to_int(b) == value
    assert oracle_np.int_to_bytes(value) == b
    assert oracle_np.bytes_to_int(b) == value


# B2 int8 / B3 string(2)  (DataType.scala:59-61, :69)
def test_b2_b3_bytes():
    col = RawColumn(DENSE_TINYINT, 1, np.array([25, -128, 127], np.int8), [3])
    assert col.dat.tobytes().hex() == "19807f"
    v = oracle_np.decode_block(col.dat, DENSE_TINYINT, 1)
    assert v.tolist() == [25, -128, 127]
    assert "CA".encode().hex() == "4341"


# B4 narrowing (Select.scala:65,73; JVM d2i, i2b)
B4 = [(18.0, 18, 18), (18.9, 18, 18), (200.0, 200, -56), (127.0, 127, 127), (128.0, 128, -128),
      (256.0, 256, 0), (3e9, 2147483647, -1), (-5.5, -5, -5), (float("nan"), 0, 0),
      (-3e9, -2147483648, 0), (1e12, 2147483647, -1), (-1e12, -2147483648, 0), (-0.9, 0, 0)]


@pytest.mark.parametrize("d,i,b", B4)
def test_b4_narrowing(d, i, b):
    assert oracle_c.d2i(d) == i and oracle_c.d2b(d) == b
    assert oracle_np.to_int(d) == i and oracle_np.to_byte(d) == b


# B5 bitmap (Select.scala:75-78,113-116)
def test_b5_bitmap():
    ages = RawColumn(DENSE_TINYINT, 1, np.array([17, 18, 19, 29, 30, 31], np.int8), [6])
    sels = [(0, GT, 18.0), (0, LT, 30.0)]
    words, count = oracle_c.scan_select([ages.ocol()], sels, 1024)
    assert count == 2 and words.tolist() == [0xC]
    w2, c2, _ = oracle_np.scan_select([ages.npcol()], sels, 1024)
    assert c2 == 2 and w2.tolist() == [0xC]


def _test100_cols():
    from immutable3_amd import synth
    t = synth.test_100()
    age = RawColumn(DENSE_TINYINT, 1, t["age"], [100])
    idc = RawColumn(DENSE_INT, 4, t["id"], [100])
    st = RawColumn(DENSE_STRING, 2, t["state"], [100])
    return age, idc, st


# B6 + B8: C1 query on the closed-form test_100 table
def test_b8_test100_range_project():
    age, idc, st = _test100_cols()
    used = [age, idc]                       # B6: scan cols for `select id, age ... where age...` = [age, id]
    sels = [(0, GT, 18.0), (0, LT, 30.0)]
    for flavour in (0, 1):
        words, count = oracle_c.scan_select([c.ocol() for c in used], sels, 1024, flavour)
        assert count == 11
        assert words.tolist() == [0x0042100108008400, 0x0000000001080084]
    w2, c2, masks = oracle_np.scan_select([c.npcol() for c in used], sels, 1024)
    assert c2 == 11 and w2.tolist() == [0x0042100108008400, 0x0000000001080084]
    ids = [10, 15, 27, 32, 44, 49, 54, 66, 71, 83, 88]
    assert np.flatnonzero(masks[0]).tolist() == ids
    expect = [(10, 21), (15, 26), (27, 20), (32, 25), (44, 19), (49, 24), (54, 29), (66, 23), (71, 28), (83, 22)]
    n, batch, pos, vals, wt = oracle_c.project([c.ocol() for c in used], [1, 0], 10, 1024, words)
    assert n == 10 and not wt
    got = list(zip(vals[0].view("<i4").reshape(-1).tolist(), vals[1].view(np.int8).reshape(-1).tolist()))
    assert got == expect
    assert pos.tolist() == ids[:10] and batch.tolist() == [0] * 10
    rows, where, wt2 = oracle_np.project([c.npcol() for c in used], [1, 0], 10, masks)
    assert rows == expect and not wt2


def test_b8_test100_match():
    age, idc, st = _test100_cols()
    words, count = oracle_c.scan_select([st.ocol()], [(0, MATCH, [b"CA"])], 1024)
    assert count == 15
    w2, c2, masks = oracle_np.scan_select([st.npcol()], [(0, MATCH, [b"CA"])], 1024)
    assert np.flatnonzero(masks[0]).tolist() == list(range(0, 100, 7))
    assert (words == w2).all()
    sels = [(0, GT, 18.0), (0, LT, 30.0), (1, MATCH, [b"CA"])]
    words, count = oracle_c.scan_select([age.ocol(), st.ocol(), idc.ocol()], sels, 1024)
    assert count == 1
    n, _, pos, vals, _ = oracle_c.project([age.ocol(), st.ocol(), idc.ocol()], [2], 0, 1024, words)
    assert n == 1 and vals[0].view("<i4").reshape(-1).tolist() == [49]


# B7 loader layout (Segment.scala:99-151, LoaderCli.scala:135-154)
B7 = [((100, 1024, 100), [[100]]), ((25, 4, 2), [[4, 4, 1], [4, 4, 1], [4, 3]]), ((8, 4, 2), [[4, 4]]), ((9, 4, 2), [[4, 4, 1]])]


@pytest.mark.parametrize("cfg,expect", B7)
def test_b7_loader_layout(tmp_path, cfg, expect):
    from immutable3_amd.schema import CodecType, Column, Table
    from immutable3_amd.storage import SegmentManager, load_rows
    rows, B, S = cfg
    t = Table("t", [Column.make("id", CodecType.DENSE_INT)], B)
    load_rows(str(tmp_path), t, [[str(i)] for i in range(rows)], S)
    sm = SegmentManager(str(tmp_path))
    got = [(np.diff(m.blockOffsets) // 4).tolist() for m in sm.segmentsMeta["t.id"]]
    assert got == expect
    allv = np.concatenate([np.asarray(d).view("<i4") for d in sm.segments["t.id"]])
    assert allv.tolist() == list(range(rows))


# error behaviour (Select.scala:22,41,80,118,156)
def test_errors():
    age, idc, st = _test100_cols()
    for cond in (NOTMATCH, NOOP):
        with pytest.raises(oracle_c.OracleError) as e:
            oracle_c.scan_select([age.ocol()], [(0, cond, None)], 1024)
        assert e.value.code == oracle_c.ERR_UNSUPPORTED_CONDITION
        with pytest.raises(oracle_np.RefException):
            oracle_np.scan_select([age.npcol()], [(0, cond, None)], 1024)
    for col, cond, operand in ((st, GT, 1.0), (st, EQ, 1.0), (age, MATCH, [b"CA"]), (idc, MATCH, [b"CA"])):
        with pytest.raises(oracle_c.OracleError) as e:
            oracle_c.scan_select([col.ocol()], [(0, cond, operand)], 1024)
        assert e.value.code == oracle_c.ERR_UNSUPPORTED_VECTOR
        with pytest.raises(oracle_np.RefException):
            oracle_np.scan_select([col.npcol()], [(0, cond, operand)], 1024)


def test_tinyint_threshold_wrap():
    """GT(200) on TINYINT means > -56; GT(3e9) means > -1 (SURVEY A.1 rule 5)."""
    vals = np.arange(-128, 128, dtype=np.int16).astype(np.int8)
    c = RawColumn(DENSE_TINYINT, 1, vals, [256])
    _, count = oracle_c.scan_select([c.ocol()], [(0, GT, 200.0)], 1024)
    assert count == int((vals > -56).sum())
    _, count = oracle_c.scan_select([c.ocol()], [(0, GT, 3e9)], 1024)
    assert count == int((vals > -1).sum())


def test_project_empty_batch_reported_not_replicated():
    """A batch with zero survivors makes the reference's ProjectIterator index out of bounds
    (Project.scala:39-57); the oracle skips it and reports would_throw."""
    v = np.array([1, 1, 1, 1, 50, 50, 50, 50], np.int32)
    c = RawColumn(DENSE_INT, 4, v, [4, 4])
    words, count = oracle_c.scan_select([c.ocol()], [(0, GT, 10.0)], 4)
    assert count == 4 and words.tolist() == [0x0, 0xF]
    n, batch, pos, vals, wt = oracle_c.project([c.ocol()], [0], 0, 4, words)
    assert n == 4 and wt and batch.tolist() == [1, 1, 1, 1] and pos.tolist() == [0, 1, 2, 3]
