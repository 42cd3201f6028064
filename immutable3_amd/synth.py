"""Seeded synthetic tables of BASELINE.json's configs.  Host-side numpy only.

    C1 test_100 : 100 rows, id = i, age = (i*37 + 11) % 90, state = CODES7[i % 7]
    C2          : int32 uniform in [0, 2^30) from splitmix64(seed=1); predicate GT(2^28) AND LT(3*2^28)
    C3          : id = i (int32), age uniform 0..99 (int8, seed=2)
    C4          : state uniform over 51 two-letter codes (seed=3)
    C5          : segment s: seed = 100 + s, id = s*10^8 + i
"""
from __future__ import annotations

import numpy as np

from .schema import CodecType, Column, Table

CODES7 = ["CA", "NY", "TX", "WA", "VA", "DC", "CT"]
CODES51 = [
    "AL", "AK", "AZ", "AR", "CA", "CO", "CT", "DE", "FL", "GA", "HI", "ID", "IL", "IN", "IA", "KS", "KY",
    "LA", "ME", "MD", "MA", "MI", "MN", "MS", "MO", "MT", "NE", "NV", "NH", "NJ", "NM", "NY", "NC", "ND",
    "OH", "OK", "OR", "PA", "RI", "SC", "SD", "TN", "TX", "UT", "VT", "VA", "WA", "WV", "WI", "WY", "DC",
]

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, n: int, start: int = 0) -> np.ndarray:
    """Outputs start .. start+n-1 of the splitmix64 stream seeded with `seed` (uint64[n])."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _chunks(n: int, step: int = 1 << 24):
    for s in range(0, n, step):
        yield s, min(step, n - s)


def uniform_int30(seed: int, n: int) -> np.ndarray:
    """C2 values: top 30 bits of splitmix64 -> int32 uniform in [0, 2^30)."""
    out = np.empty(n, dtype=np.int32)
    for s, m in _chunks(n):
        out[s: s + m] = (splitmix64(seed, m, s) >> np.uint64(34)).astype(np.int32)
    return out


def uniform_below(seed: int, n: int, k: int, dtype=np.int32) -> np.ndarray:
    """floor(u32 * k / 2^32) with u32 the top 32 bits of splitmix64: uniform in [0, k)."""
    out = np.empty(n, dtype=dtype)
    for s, m in _chunks(n):
        hi = splitmix64(seed, m, s) >> np.uint64(32)
        out[s: s + m] = ((hi * np.uint64(k)) >> np.uint64(32)).astype(dtype)
    return out


def state_codes(seed: int, n: int, codes=CODES51) -> np.ndarray:
    """uint8[n, 2] of two-letter codes drawn uniformly from `codes`."""
    table = np.array([list(c.encode("ascii")) for c in codes], dtype=np.uint8)
    return table[uniform_below(seed, n, len(codes), np.int32)]


def table_schema(name: str, block_size: int = 1024) -> Table:
    """id:DENSE_INT, state:DENSE_STRING:size=2, age:DENSE_TINYINT  (README.md:10 of the reference)."""
    return Table(name, [
        Column.make("id", CodecType.DENSE_INT),
        Column.make("state", CodecType.DENSE_STRING, {"size": "2"}),
        Column.make("age", CodecType.DENSE_TINYINT),
    ], block_size)


def test_100():
    """C1: closed-form 100-row table."""
    i = np.arange(100, dtype=np.int64)
    return {
        "id": i.astype(np.int32),
        "age": ((i * 37 + 11) % 90).astype(np.int8),
        "state": np.array([list(CODES7[k % 7].encode("ascii")) for k in range(100)], dtype=np.uint8),
    }


def c3_segment(n: int, seed: int = 2, id_base: int = 0):
    """C3 / C5 columns: id = id_base + i, age uniform 0..99."""
    return {
        "id": (np.arange(n, dtype=np.int64) + id_base).astype(np.int32),
        "age": uniform_below(seed, n, 100, np.int8),
    }


def block_offsets(n_rows: int, width: int, block_size: int = 1024) -> np.ndarray:
    """blockOffset table of a column cut into block_size-row blocks (short last block)."""
    full, rem = divmod(n_rows, block_size)
    rows = np.array([block_size] * full + ([rem] if rem else []), dtype=np.int64)
    return np.concatenate([[0], np.cumsum(rows * width)]).astype(np.int32)
