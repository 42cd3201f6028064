"""Iteration order of a Scala 2.12 immutable Set[Column] -- what decides Engine.getColumns' column order
(engine/src/main/scala/immutabledb/engine/Engine.scala:105: `(rec(query.select).toList ++ projectColumns).toSet.toList`).

The algorithm lives in scala-library 2.12.11 (build.sbt:2), a dependency that is not under /root/reference and cannot run
here (no JVM): it is RESTATED from its published source --
  immutable.Set.Set1..Set4          keep insertion order; Set4 + e = new HashSet + (e1, e2, e3, e4, e)
  immutable.HashSet                 a 32-way hash trie over improve(elem.##); iteration walks each node's children in
                                    ascending index = ascending 5-bit chunks of the improved hash, LOW bits first
  HashSet.improve(h)                h += ~(h << 9); h ^= h >>> 14; h += h << 4; h ^= h >>> 10
  case class hashCode               MurmurHash3.productHash(x, 0xcafebabe): mix over the fields' ##, finalizeHash(h, arity)
  Enumeration#Value.hashCode        id
  immutable.Map.hashCode            MurmurHash3.unorderedHash(entries, "Map".hashCode); an entry is a Tuple2 (a case class)
  String.##                         java.lang.String.hashCode
PARITY UNPINNED at this boundary: the reference holds no test for it.  Pins used: MurmurHash3's mix / finalizeHash are the
standard x86_32 block and finaliser and are checked against an independent implementation (sklearn's murmurhash3_32,
tests/test_host.py); the composition above is from the library's source as published.  With <= 4 distinct columns -- every
query over the reference's own 3-column tables -- none of this is reached."""
from __future__ import annotations

from typing import Dict, List, Sequence

M32 = 0xFFFFFFFF


def _i32(x: int) -> int:
    x &= M32
    return x - (1 << 32) if x & 0x80000000 else x


def java_string_hash(s: str) -> int:
    """java.lang.String.hashCode: s[0]*31^(n-1) + ... over UTF-16 code units, 32-bit wrap-around."""
    raw = s.encode("utf-16-be")
    h = 0
    for i in range(0, len(raw), 2):
        h = (31 * h + int.from_bytes(raw[i:i + 2], "big")) & M32
    return _i32(h)


def _rotl(x: int, r: int) -> int:
    x &= M32
    return ((x << r) | (x >> (32 - r))) & M32


def mix_last(h: int, data: int) -> int:
    k = (data & M32) * 0xcc9e2d51 & M32
    k = _rotl(k, 15)
    k = k * 0x1b873593 & M32
    return (h ^ k) & M32


def mix(h: int, data: int) -> int:
    h = mix_last(h, data)
    h = _rotl(h, 13)
    return (h * 5 + 0xe6546b64) & M32


def finalize_hash(h: int, length: int) -> int:
    h = (h ^ length) & M32
    h ^= h >> 16
    h = h * 0x85ebca6b & M32
    h ^= h >> 13
    h = h * 0xc2b2ae35 & M32
    h ^= h >> 16
    return h & M32


PRODUCT_SEED = 0xcafebabe
MAP_SEED = java_string_hash("Map") & M32


def product_hash(field_hashes: Sequence[int]) -> int:
    """MurmurHash3.productHash(x) of a case class with these field ## values (arity >= 1)."""
    h = PRODUCT_SEED
    for f in field_hashes:
        h = mix(h, f & M32)
    return finalize_hash(h, len(field_hashes))


def map_hash(entries: Dict[str, str]) -> int:
    """immutable.Map[String, String].hashCode = MurmurHash3.unorderedHash(tuples, mapSeed)."""
    a = b = n = 0
    c = 1
    for k, v in entries.items():
        h = product_hash([java_string_hash(k), java_string_hash(v)])
        a = (a + h) & M32
        b ^= h
        if h != 0:
            c = c * h & M32
        n += 1
    h = MAP_SEED
    h = mix(h, a)
    h = mix(h, b)
    h = mix_last(h, c)
    return finalize_hash(h, n)


def improve(hcode: int) -> int:
    h = (hcode + (~((hcode << 9) & M32) & M32)) & M32
    h ^= h >> 14
    h = (h + ((h << 4) & M32)) & M32
    return (h ^ (h >> 10)) & M32


COLUMN_TYPE_ID = {"INT": 0, "TINYINT": 1, "STRING": 2}                                  # core/Column.scala:13-16
CODEC_ID = {"PFOR_INT": 0, "DENSE_INT": 1, "DENSE_TINYINT": 2, "DENSE_STRING": 3}       # core/codec/Codec.scala:21-24


def column_hash(name: str, column_type: str, codec: str, dtype_attrs: Dict[str, str]) -> int:
    """Column.hashCode (case class Column(name, columnType, codec, dtypeAttrs), core/Column.scala:18)."""
    return product_hash([java_string_hash(name), COLUMN_TYPE_ID[column_type], CODEC_ID[codec], map_hash(dtype_attrs)])


def trie_key(hcode: int):
    """Sort key that reproduces HashTrieSet iteration: 5-bit chunks of improve(hcode), low bits first."""
    h = improve(hcode & M32)
    return tuple((h >> s) & 31 for s in range(0, 35, 5))


class ScalaSet:
    """immutable.Set[A] as far as its ITERATION ORDER goes: elements are (value, hashCode) pairs."""

    def __init__(self):
        self.small: List = []      # Set1..Set4: insertion order
        self.trie = None           # HashSet: dict value -> hash

    def add(self, value, hcode: int):
        if self.trie is not None:
            self.trie.setdefault(value, hcode)
            return self
        if any(v == value for v, _ in self.small):
            return self
        if len(self.small) < 4:
            self.small.append((value, hcode))
        else:                       # Set4 + elem: new HashSet + (elem1, elem2, elem3, elem4, elem)
            self.trie = {v: h for v, h in self.small}
            self.trie[value] = hcode
            self.small = []
        return self

    def to_list(self) -> List:
        if self.trie is None:
            return [v for v, _ in self.small]
        return [v for v, h in sorted(self.trie.items(), key=lambda kv: trie_key(kv[1]))]
