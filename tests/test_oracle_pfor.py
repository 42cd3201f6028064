"""PFOR_INT block codec: the oracle's restatement of the format the reference's ENCODER writes
(core/codec/PFORCodec.scala:19-31 over JavaFastPFOR 0.1.10's IntegratedIntCompressor).

Parity unpinned by the reference (its decode throws, it has no tests, the library is not available offline):
pinned here by hand-derived known-answer blocks, by two independent restatements (C word-by-word, numpy bit-stream)
agreeing byte for byte, and by encode -> decode round trips over every structural case."""
import numpy as np
import pytest

from oracle import oracle_np


def be(words):
    return b"".join(int(w & 0xFFFFFFFF).to_bytes(4, "big") for w in words)


# ---- hand-derived known answers ---------------------------------------------------------------------
def test_kat_sequential_128(oracle):
    # 0..127: deltas 0,1,1,... -> every mini-block has width 1; first word of the group header, then 4 x 1 word.
    # mini-block 0: delta bits (0,1,1,...,1) LSB-first = 0xFFFFFFFE; the others 0xFFFFFFFF.
    want = be([128, 0x01010101, 0xFFFFFFFE, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF]) + b"\0" * 8
    v = np.arange(128, dtype=np.int32)
    assert oracle.pfor_encode_block(v) == want
    assert oracle_np.pfor_encode_block(v) == want
    assert oracle.pfor_decode_block(want).tolist() == v.tolist()


def test_kat_variable_byte_tail(oracle):
    # 3 values < one mini-block: all variable-byte.  deltas 5, 295, 0 -> 0x85 | 0x27 0x82 | 0x80; packed little-endian
    # into 0x80822785, written big-endian.
    want = be([3, 0x80822785]) + b"\0" * 8
    v = np.array([5, 300, 300], dtype=np.int32)
    assert oracle.pfor_encode_block(v) == want
    assert oracle_np.pfor_encode_block(v) == want
    assert oracle.pfor_decode_block(want).tolist() == [5, 300, 300]


def test_kat_leftover_miniblock_and_raw(oracle):
    # 32 values = one leftover mini-block (own header word).  Values 7 then 31 x 6: the second delta is -1 -> the OR
    # of the deltas has bit 31 set -> width 32 -> the VALUES are stored, not deltas (integratedpack32 = arraycopy).
    v = np.array([7] + [6] * 31, dtype=np.int32)
    want = be([32, 32] + v.tolist()) + b"\0" * 8
    assert oracle.pfor_encode_block(v) == want
    assert oracle_np.pfor_encode_block(v) == want
    # constant run: first delta 9 - 0 = 9 -> width 4; word 0 = 9 (value 0 in bits 0..3), three zero words
    v = np.full(32, 9, dtype=np.int32)
    want = be([32, 4, 9, 0, 0, 0]) + b"\0" * 8
    assert oracle.pfor_encode_block(v) == want
    # all zeros: width 0 -> header only
    assert oracle.pfor_encode_block(np.zeros(32, dtype=np.int32)) == be([32, 0]) + b"\0" * 8
    # 160 values = one group of four + one leftover + nothing else
    v = np.zeros(160, dtype=np.int32)
    assert oracle.pfor_encode_block(v) == be([160, 0, 0]) + b"\0" * 8


def test_kat_straddling_width(oracle):
    # width 3: value i occupies bits [3i, 3i+3); 32 values of delta 5 (0b101) -> 3 words of the repeating pattern
    v = (np.arange(1, 33) * 5).astype(np.int32)
    stream = 0
    for i in range(32):
        stream |= 5 << (3 * i)
    words = [(stream >> (32 * k)) & 0xFFFFFFFF for k in range(3)]
    want = be([32, 3] + words) + b"\0" * 8
    assert oracle.pfor_encode_block(v) == want
    assert oracle.pfor_decode_block(want).tolist() == v.tolist()


def test_negative_first_delta_wraps(oracle):
    # deltas wrap in 32 bits: a negative first value makes the first delta's top bit set -> width 32 for that mini-block
    v = np.arange(-5, 27, dtype=np.int32)
    e = oracle.pfor_encode_block(v)
    assert e[:8] == be([32, 32])
    assert oracle.pfor_decode_block(e).tolist() == v.tolist()


# ---- two restatements agree, and round-trip ------------------------------------------------------------
SIZES = [1, 2, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 159, 160, 161, 255, 256, 1000, 1023, 1024, 1025, 2048, 4096 + 77]


@pytest.mark.parametrize("n", SIZES)
def test_cross_and_roundtrip(oracle, n):
    rng = np.random.default_rng(n)
    cases = [
        np.sort(rng.integers(-2**31, 2**31, n)),          # sorted, wide deltas
        rng.integers(-2**31, 2**31, n),                   # unsorted: raw mini-blocks
        np.cumsum(rng.integers(0, 5, n)),                 # small deltas
        np.full(n, 7),                                    # width 0 after the first mini-block
        np.cumsum(rng.integers(0, 2**20, n)) - 2**30,     # medium deltas, negative start
        np.where(rng.random(n) < 0.02, -1, 1).cumsum(),   # mostly sorted with rare negative deltas (mixed raw / packed)
        np.arange(n) * 3 - 50,
    ]
    for v in cases:
        v = v.astype(np.int64).astype(np.int32)
        e = oracle.pfor_encode_block(v)
        assert e == oracle_np.pfor_encode_block(v)
        assert len(e) % 4 == 0 and e[-8:] == b"\0" * 8 and int.from_bytes(e[:4], "big") == n
        assert oracle.pfor_decode_block(e).tolist() == v.tolist()
        assert oracle_np.pfor_decode_block(e).tolist() == v.tolist()


def test_malformed_blocks_are_refused(oracle):
    e = bytearray(oracle.pfor_encode_block(np.arange(128, dtype=np.int32)))
    bad = bytes(e[:4]) + (40 << 24).to_bytes(4, "big") + bytes(e[8:])  # a width above 32
    with pytest.raises(oracle.OracleError):
        oracle.pfor_decode_block(bad)
    with pytest.raises(oracle.OracleError):
        oracle.pfor_decode_block(bytes(e[:12]))                          # data runs past the block
    with pytest.raises(oracle.OracleError):
        oracle.pfor_decode_block(b"\0\0")                                # no count word


def test_column_helpers(oracle):
    v = np.arange(5000, dtype=np.int32) * 2
    dat, offs = oracle.pfor_encode_column(v, 1024)
    assert len(offs) == 6 and offs[0] == 0 and offs[-1] == dat.size
    assert oracle.pfor_decode_column(dat, offs).tolist() == v.tolist()
