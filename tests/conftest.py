import os
import sys

import numpy as np
import pytest

# PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so); libimm3.so links the system one.  Whichever is
# loaded FIRST serves the whole process -- a second runtime instance finds no GPU ("No HIP GPUs are available").  Tests
# that use torch for device memory next to the library therefore need torch loaded before the first native.load(),
# whatever order the test files run in: load it here, once, for every session.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- helpers shared by the oracle tests and the GPU parity tests --------------------------------
PFOR_INT, DENSE_INT, DENSE_TINYINT, DENSE_STRING = 0, 1, 2, 3
MATCH, NOTMATCH, EQ, GT, LT, NOOP = 0, 1, 2, 3, 4, 5


def offsets_for(block_rows, width):
    return np.concatenate([[0], np.cumsum(np.array(block_rows, dtype=np.int64) * width)]).astype(np.int32)


def blocks_of(n, block_size):
    full, rem = divmod(n, block_size)
    return [block_size] * full + ([rem] if rem else [])


class RawColumn:
    """(codec, width, raw bytes, block offsets) of one column of one segment."""

    def __init__(self, codec, width, values, block_rows):
        self.codec, self.width = codec, width
        if codec == DENSE_INT:
            self.dat = np.ascontiguousarray(values, dtype="<i4").view(np.uint8).copy()
        elif codec == DENSE_TINYINT:
            self.dat = np.ascontiguousarray(values, dtype=np.int8).view(np.uint8).copy()
        else:
            self.dat = np.ascontiguousarray(values, dtype=np.uint8).reshape(-1).copy()
        self.offsets = offsets_for(block_rows, width)
        self.values = values

    def ocol(self):
        from oracle import oracle_c
        return oracle_c.OColumn(self.dat, self.offsets, self.codec, self.width)

    def npcol(self):
        return (self.dat, self.offsets, self.codec, self.width)

    def native(self):
        return (self.codec, self.width, self.dat, self.dat.size, self.offsets)


class PforColumn:
    """A PFOR_INT column: blocks as PFORCodecInt.encode writes them (oracle encoder).  The oracle side sees the
    DENSE_INT column holding the same values in the same blocks -- the decode the encoder implies."""

    def __init__(self, values, block_rows):
        from oracle import oracle_c
        self.codec, self.width = PFOR_INT, 4
        self.values = np.ascontiguousarray(values, dtype=np.int32)
        self.block_rows = list(block_rows)
        parts, offs, pos = [], [0], 0
        for n in self.block_rows:
            blk = oracle_c.pfor_encode_block(self.values[pos:pos + n])   # (an empty block is the 12 bytes the reference's encoder writes for no values)
            parts.append(blk)
            offs.append(offs[-1] + len(blk))
            pos += n
        self.dat = np.frombuffer(b"".join(parts) or b"", dtype=np.uint8).copy()
        self.offsets = np.array(offs, dtype=np.int32)
        self._dense = RawColumn(DENSE_INT, 4, self.values, self.block_rows)

    def ocol(self):
        return self._dense.ocol()

    def npcol(self):
        return self._dense.npcol()

    def native(self):
        return (self.codec, self.width, self.dat, self.dat.size, self.offsets)


SNAPPY_INT, SNAPPY_TINYINT, SNAPPY_STRING = 16, 17, 18


class SnappyColumn:
    """A snappy-coded column: every block is what SnappyCodec.encode writes for its raw value bytes (oracle encoder, or
    -- encoder="google" -- the stream framing built here around pyarrow's Google-snappy payload, chunked at 32768 input
    bytes).  The oracle side sees the DENSE_* column with the same values in the same blocks."""

    def __init__(self, dense_codec, width, values, block_rows, encoder="oracle"):
        from oracle import oracle_c
        self.codec = {DENSE_INT: SNAPPY_INT, DENSE_TINYINT: SNAPPY_TINYINT, DENSE_STRING: SNAPPY_STRING}[dense_codec]
        self.width = width
        self._dense = RawColumn(dense_codec, width, values, block_rows)
        raw = self._dense.dat
        parts, offs, pos = [], [0], 0
        for n in block_rows:
            chunk = raw[pos * width:(pos + n) * width].tobytes()
            if encoder == "oracle":
                blk = oracle_c.snappy_block_encode(chunk)
            else:
                import pyarrow as pa
                codec = pa.Codec("snappy")
                blk = b"snappy\x00"
                for s in range(0, len(chunk), 32768):
                    piece = chunk[s:s + 32768]
                    payload = codec.compress(piece, asbytes=True)
                    blk += bytes([1]) + len(payload).to_bytes(2, "big") + oracle_c.crc32c_masked(piece).to_bytes(4, "big") + payload
            parts.append(blk)
            offs.append(offs[-1] + len(blk))
            pos += n
        self.dat = np.frombuffer(b"".join(parts) or b"", dtype=np.uint8).copy()
        self.offsets = np.array(offs, dtype=np.int32)
        self.values = values

    def ocol(self):
        return self._dense.ocol()

    def npcol(self):
        return self._dense.npcol()

    def native(self):
        return (self.codec, self.width, self.dat, self.dat.size, self.offsets)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_c
    oracle_c.build()
    return oracle_c
