"""Snappy-coded blocks: the oracle's restatement of the format SnappyCodec.encode writes (core/codec/SnappyCodec.scala:
14-43: iq80 snappy 0.4 SnappyOutputStream framing around raw Snappy).  The reference cannot read this format (`decode =
???`, no CodecType), so parity is unpinned at the reference boundary; pins: hand-made known-answer streams, the CRC-32C
check value, two independent restatements, and -- for the raw Snappy layer -- interoperability in both directions with
pyarrow's bundled Google snappy."""
import numpy as np
import pytest

from oracle import oracle_np


def test_crc32c_check_values(oracle):
    assert oracle.crc32c(b"123456789") == 0xE3069283 == oracle_np.crc32c(b"123456789")     # the standard CRC-32C check value
    assert oracle.crc32c(b"") == 0 and oracle.crc32c(bytes(32)) == 0x8A9136AA == oracle_np.crc32c(bytes(32))  # RFC 3720 B.4
    c = 0xE3069283
    assert oracle.crc32c_masked(b"123456789") == (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF == oracle_np.crc32c_masked(b"123456789")


def test_kat_raw_elements(oracle):
    # preamble 11; literal "abc" (tag (3-1)<<2); copy len 4 offset 3 via tag-01 ((4-4)<<2 | 1, offset byte 3): overlapping
    # -> "abca"; copy len 4 offset 7 via tag-10 ((4-1)<<2 | 2, 07 00) -> "abca"
    raw = bytes([11, 0x08]) + b"abc" + bytes([0x01, 0x03, 0x0E, 0x07, 0x00])
    assert oracle.snappy_raw_decode(raw) == b"abcabcaabca" == oracle_np.snappy_raw_decode(raw)
    # 61-byte literal needs one length byte: tag 60<<2, then 60 (= len - 1)
    raw = bytes([61, 60 << 2, 60]) + bytes(range(61))
    assert oracle.snappy_raw_decode(raw) == bytes(range(61)) == oracle_np.snappy_raw_decode(raw)
    # run-length: literal "x" then copy len 64 offset 1 (tag-10: (64-1)<<2 | 2)
    raw = bytes([65, 0x00]) + b"x" + bytes([(63 << 2) | 2, 1, 0])
    assert oracle.snappy_raw_decode(raw) == b"x" * 65 == oracle_np.snappy_raw_decode(raw)
    # 4-byte-offset copy (tag-11)
    raw = bytes([8, 0x0C]) + b"wxyz" + bytes([(3 << 2) | 3, 4, 0, 0, 0])
    assert oracle.snappy_raw_decode(raw) == b"wxyzwxyz" == oracle_np.snappy_raw_decode(raw)


def test_kat_stream_framing(oracle):
    data = b"hello hello hello hello"
    crc = oracle_np.crc32c_masked(data)
    stored = b"snappy\x00" + bytes([0, 0, len(data)]) + crc.to_bytes(4, "big") + data
    assert oracle.snappy_block_decode(stored) == data == oracle_np.snappy_block_decode(stored)
    raw = bytes([len(data), (6 - 1) << 2]) + b"hello " + bytes([((17 - 1) << 2) | 2, 6, 0])
    comp = b"snappy\x00" + bytes([1, 0, len(raw)]) + crc.to_bytes(4, "big") + raw
    assert oracle.snappy_block_decode(comp) == data == oracle_np.snappy_block_decode(comp)
    # what the encoder writes for it: compressed (11/23 <= 7/8), one chunk
    enc = oracle.snappy_block_encode(data)
    assert enc[:8] == b"snappy\x00\x01" and enc[10:14] == crc.to_bytes(4, "big") and oracle.snappy_block_decode(enc) == data
    # incompressible input is stored (flag 0), inputs above 32768 bytes are cut into chunks
    rnd = np.random.default_rng(1).integers(0, 256, 40000, dtype=np.uint8).tobytes()
    enc = oracle.snappy_block_encode(rnd)
    assert enc[7] == 0 and int.from_bytes(enc[8:10], "big") == 32768
    second = 7 + 7 + 32768
    assert enc[second] == 0 and int.from_bytes(enc[second + 1:second + 3], "big") == 40000 - 32768
    assert oracle.snappy_block_decode(enc) == rnd == oracle_np.snappy_block_decode(enc)
    # empty block: header only
    assert oracle.snappy_block_encode(b"") == b"snappy\x00" and oracle.snappy_block_decode(b"snappy\x00") == b""


def test_malformed_is_refused(oracle):
    data = b"hello hello hello hello"
    enc = bytearray(oracle.snappy_block_encode(data))
    for mutate in (lambda b: b.__setitem__(0, ord("S")),          # bad stream header
                   lambda b: b.__setitem__(7, 2),                 # unknown flag
                   lambda b: b.__setitem__(12, b[12] ^ 1),        # checksum mismatch
                   lambda b: b.__setitem__(9, b[9] + 5),          # payload longer than the block
                   lambda b: b.__setitem__(len(b) - 2, 0xFF)):    # copy offset beyond the output
        bad = bytearray(enc)
        mutate(bad)
        with pytest.raises(oracle.OracleError):
            oracle.snappy_block_decode(bytes(bad))


@pytest.mark.parametrize("n", [0, 1, 5, 60, 61, 100, 255, 256, 257, 4096, 32768, 32769, 70000])
def test_interop_with_google_snappy_and_roundtrips(oracle, n):
    pa = pytest.importorskip("pyarrow")
    codec = pa.Codec("snappy")
    rng = np.random.default_rng(n)
    cases = [rng.integers(0, 256, n, dtype=np.uint8).tobytes(), (np.arange(n) % 7).astype(np.uint8).tobytes(), bytes(n),
             np.repeat(rng.integers(0, 50, max(1, n // 16) + 1, dtype=np.uint8), 16)[:n].tobytes(),
             np.arange(n // 4 + 1, dtype="<i4").tobytes()[:n], b"CANYTXWA"[: max(1, n % 9)] * (n // max(1, n % 9) + 1)]
    for d in cases:
        d = d[:n]
        ours = oracle.snappy_raw_encode(d)
        assert oracle.snappy_raw_decode(ours) == d == oracle_np.snappy_raw_decode(ours)
        if n:
            assert codec.decompress(ours, decompressed_size=len(d), asbytes=True) == d     # Google's decoder reads our stream
        theirs = codec.compress(d, asbytes=True)
        assert oracle.snappy_raw_decode(theirs) == d == oracle_np.snappy_raw_decode(theirs)  # we read Google's stream
        blk = oracle.snappy_block_encode(d)
        assert oracle.snappy_block_decode(blk) == d == oracle_np.snappy_block_decode(blk)
