"""On-disk format of the reference: reader (SegmentManager / Segment) and writer (SegmentWriter / loader).

Mirrors core/src/main/scala/immutabledb/storage/{Segment,SegmentManager}.scala and
loader/src/main/scala/immutabledb/loader/LoaderCli.scala:113-154, including the loader's quirk that a
"full" segment holds segmentSize*blockSize + 1 rows in segmentSize + 1 blocks.

    <dataDir>/<table>/_table.meta      schema (schema.TableIO)
    <dataDir>/<table>/<col>_<id>.dat   concatenated encoded blocks (DENSE_* => raw little-endian values)
    <dataDir>/<table>/<col>_<id>.meta  {"blockOffset":[0, e1, ..., eN]}   byte end offsets
"""
from __future__ import annotations

import json
import os
import re
from dataclasses import dataclass
from typing import Dict, Iterator, List, Sequence

import numpy as np

from .schema import CodecType, Column, Table, TableIO

_INT_RE = re.compile(r"^[+-]?\d+$")


# ------------------------------------------------------------------------------------------
# DataType.stringToValue / valueToBytes (core/.../DataType.scala:31-71)
# ------------------------------------------------------------------------------------------
def string_to_bytes(col: Column, s: str) -> bytes:
    if col.codec in CodecType.INT_CODECS:
        if not _INT_RE.match(s) or not (-(2 ** 31) <= int(s) <= 2 ** 31 - 1):
            raise ValueError(f'NumberFormatException: For input string: "{s}"')  # s.toInt
        return int(s).to_bytes(4, "little", signed=True)                          # IntType.valueToBytes :40-47
    if col.codec in CodecType.TINYINT_CODECS:
        if not _INT_RE.match(s) or not (-128 <= int(s) <= 127):
            raise ValueError(f'NumberFormatException: Value out of range. Value:"{s}" Radix:10')  # s.toByte
        return int(s).to_bytes(1, "little", signed=True)
    if col.codec in CodecType.STRING_CODECS:
        return s.encode("utf-8")   # value.getBytes(): NO padding / truncation to dtypeAttrs("size") (:69)
    raise Exception("")


@dataclass
class SegmentMeta:                 # Segment.scala:33
    blockOffsets: np.ndarray       # int32, N+1 entries, first 0

    @staticmethod
    def load(path: str) -> "SegmentMeta":    # Segment.scala:35-50 (key is singular: "blockOffset")
        with open(path) as f:
            j = json.load(f)
        return SegmentMeta(np.array([int(x) for x in j["blockOffset"]], dtype=np.int32))

    @staticmethod
    def store(path: str, meta: "SegmentMeta"):
        with open(path, "w") as f:
            json.dump({"blockOffset": [int(x) for x in meta.blockOffsets]}, f, separators=(",", ":"))


class SegmentWriter:
    """Segment.scala:70-152.  write() buffers blockSize records; the (blockSize+1)-th write flushes first."""

    def __init__(self, id: int, blockSize: int, tableName: str, column: Column, dataDir: str, segmentSize: int):
        self.id, self.blockSize, self.tableName, self.column = id, blockSize, tableName, column
        self.dataDir, self.segmentSize = dataDir, segmentSize
        base = os.path.join(dataDir, tableName)
        os.makedirs(base, exist_ok=True)
        self._dat_path = os.path.join(base, f"{column.name}_{id}.dat")
        self._meta_path = os.path.join(base, f"{column.name}_{id}.meta")
        self._file = open(self._dat_path, "wb")                  # setLength(0): truncate
        self._capacity = blockSize * column.width                # ByteBuffer.allocateDirect(blockSize * dtype.size)
        self._buf = bytearray()
        self._records = 0
        self.blockBufferOffsets: List[int] = [0]

    def newSegment(self) -> "SegmentWriter":
        return SegmentWriter(self.id + 1, self.blockSize, self.tableName, self.column, self.dataDir, self.segmentSize)

    def _put(self, x: str):
        b = string_to_bytes(self.column, x)
        if len(self._buf) + len(b) > self._capacity:
            raise OverflowError("BufferOverflowException")     # blockBuffer.put past capacity
        self._buf += b
        self._records += 1

    def write(self, x: str):
        if len(self.blockBufferOffsets) > self.segmentSize:
            raise Exception("Segment full")
        if self._records < self.blockSize:
            self._put(x)
        else:
            self.flush()
            self._put(x)

    def flush(self):
        if self.column.codec == CodecType.PFOR_INT:              # PFORCodecInt.encode (PFORCodec.scala:19-31), host code of libimm3
            from . import native
            encoded = native.pfor_encode_block(np.frombuffer(bytes(self._buf), dtype="<i4"))
        elif self.column.codec in CodecType.SNAPPY:              # SnappyCodec.encode (SnappyCodec.scala:15-27), host code of libimm3
            from . import native
            encoded = native.snappy_encode_block(bytes(self._buf))
        else:
            encoded = bytes(self._buf)                           # DenseCodec.encode(bytes) is the identity (DenseCodec.scala:18-22)
        self._file.write(encoded)
        self.blockBufferOffsets.append(self.blockBufferOffsets[-1] + len(encoded))
        self._buf = bytearray()
        self._records = 0

    @property
    def remaining(self) -> int:
        return self.segmentSize - (len(self.blockBufferOffsets) - 1)

    def close(self):
        if len(self._buf) > 0:
            self.flush()
        SegmentMeta.store(self._meta_path, SegmentMeta(np.array(self.blockBufferOffsets, dtype=np.int32)))
        self._file.close()


def load_rows(dataDir: str, table: Table, rows: Sequence[Sequence[str]], segmentSize: int):
    """LoaderCli.main's load loop (LoaderCli.scala:130-154) over already-split, trimmed fields."""
    TableIO.clear(dataDir, table)
    TableIO.store(dataDir, table)
    cols = list(table.columns)
    segs: Dict[str, SegmentWriter] = {c.name: SegmentWriter(0, table.blockSize, table.name, c, dataDir, segmentSize) for c in cols}
    for vals in rows:
        for idx in range(len(vals)):
            name = cols[idx].name
            seg = segs[name]
            if seg.remaining > 0:
                seg.write(vals[idx])
            else:
                seg.close()
                segs[name] = seg.newSegment()
                segs[name].write(vals[idx])
    for seg in segs.values():
        seg.close()


def java_split_comma(line: str):
    """`line.split(",")` as java.lang.String.split does it (LoaderCli.scala:136): trailing empty strings are dropped
    BEFORE the fields are trimmed ("1,CA," -> ["1", "CA"]; ",," -> []), a line without a comma is returned whole
    ("" -> [""]).  Python's str.split keeps the trailing empties."""
    vals = line.split(",")
    if len(vals) > 1:
        while vals and vals[-1] == "":
            vals.pop()
    return vals


def load_csv(dataDir: str, table: Table, csv_path: str, segmentSize: int):
    """LoaderCli: the first line is a header and is skipped (:115-116); fields split on ',' and trimmed (:136)."""
    def gen():
        with open(csv_path) as f:
            next(f, None)
            for line in f:
                line = line.rstrip("\n").rstrip("\r")
                yield [v.strip() for v in java_split_comma(line)]
    load_rows(dataDir, table, gen(), segmentSize)


def write_segment_arrays(dataDir: str, table: Table, seg_id: int, arrays: Dict[str, np.ndarray], block_rows: Sequence[int] | None = None):
    """Bulk writer for synthetic tables: writes one segment of every column straight from numpy arrays
    (int32 / int8 / uint8[n, size]) in the reference's format.  block_rows gives the rows of each block
    (default: table.blockSize-row blocks with a short last one), so ragged layouts can be produced."""
    base = os.path.join(dataDir, table.name)
    os.makedirs(base, exist_ok=True)
    for c in table.columns:
        a = arrays[c.name]
        n = a.shape[0]
        if c.codec in CodecType.INT_CODECS:
            raw = np.ascontiguousarray(a, dtype="<i4").view(np.uint8)
        elif c.codec in CodecType.TINYINT_CODECS:
            raw = np.ascontiguousarray(a, dtype=np.int8).view(np.uint8)
        else:
            raw = np.ascontiguousarray(a, dtype=np.uint8).reshape(n, c.width).reshape(-1)
        if block_rows is None:
            full, rem = divmod(n, table.blockSize)
            br = [table.blockSize] * full + ([rem] if rem else [])
        else:
            br = list(block_rows)
            assert sum(br) == n
        if c.codec == CodecType.PFOR_INT:   # each block through PFORCodecInt.encode, as SegmentWriter.flush does
            from . import native
            vals = np.ascontiguousarray(a, dtype="<i4")
            if block_rows is None:
                raw, offs = native.pfor_encode_column(vals, table.blockSize)
            else:
                parts, offs, pos = [], [0], 0
                for r in br:
                    parts.append(native.pfor_encode_block(vals[pos:pos + r]))
                    offs.append(offs[-1] + len(parts[-1]))
                    pos += r
                raw, offs = np.frombuffer(b"".join(parts), dtype=np.uint8), np.array(offs, dtype=np.int32)
            raw.tofile(os.path.join(base, f"{c.name}_{seg_id}.dat"))
            SegmentMeta.store(os.path.join(base, f"{c.name}_{seg_id}.meta"), SegmentMeta(offs))
            continue
        if c.codec in CodecType.SNAPPY:     # each block through SnappyCodec.encode
            from . import native
            parts, offs, pos = [], [0], 0
            for r in br:
                parts.append(native.snappy_encode_block(raw[pos * c.width:(pos + r) * c.width]))
                offs.append(offs[-1] + len(parts[-1]))
                pos += r
            np.frombuffer(b"".join(parts), dtype=np.uint8).tofile(os.path.join(base, f"{c.name}_{seg_id}.dat"))
            SegmentMeta.store(os.path.join(base, f"{c.name}_{seg_id}.meta"), SegmentMeta(np.array(offs, dtype=np.int32)))
            continue
        offs = np.concatenate([[0], np.cumsum(np.array(br, dtype=np.int64) * c.width)]).astype(np.int32)
        raw.tofile(os.path.join(base, f"{c.name}_{seg_id}.dat"))
        SegmentMeta.store(os.path.join(base, f"{c.name}_{seg_id}.meta"), SegmentMeta(offs))


class Segment:
    """Segment.scala:154-181: the mmap'd buffer + block offsets; iterating yields one byte block per next()."""

    def __init__(self, id: int, segmentData: np.ndarray, meta: SegmentMeta):
        self.id, self.segmentData, self.meta = id, segmentData, meta

    def __iter__(self) -> Iterator[np.ndarray]:
        # BlockIterator: relative gets from a rewound buffer == a running cursor (Segment.scala:159-168)
        cursor = 0
        offs = self.meta.blockOffsets
        for k in range(len(offs) - 1):
            ln = int(offs[k + 1]) - int(offs[k])
            yield self.segmentData[cursor: cursor + ln]
            cursor += ln

    iterator = __iter__


class SegmentManager:
    """SegmentManager.scala:20-111: discovers tables, mmaps every <col>_*.dat, loads every .meta.
    Segment order is the LEXICOGRAPHIC filename order (:38-42, :61-65), so `_10` sorts before `_2`."""

    def __init__(self, dataDir: str):
        self.dataDir = dataDir
        dirs = sorted(d for d in os.listdir(dataDir) if os.path.isdir(os.path.join(dataDir, d)))
        self.tables: List[Table] = [TableIO.load(dataDir, d) for d in dirs]
        self.segments: Dict[str, List[np.ndarray]] = {}
        self.segmentsMeta: Dict[str, List[SegmentMeta]] = {}
        for t in self.tables:
            for c in t.columns:
                files = sorted(os.listdir(os.path.join(dataDir, t.name)))
                dats = [f for f in files if f.startswith(f"{c.name}_") and f.endswith(".dat")]
                metas = [f for f in files if f.startswith(f"{c.name}_") and f.endswith(".meta")]
                key = f"{t.name}.{c.name}"
                self.segments[key] = [self._get_byte_buffer(os.path.join(dataDir, t.name, f)) for f in dats]
                self.segmentsMeta[key] = [SegmentMeta.load(os.path.join(dataDir, t.name, f)) for f in metas]

    @staticmethod
    def _get_byte_buffer(path: str) -> np.ndarray:   # getByteBuffer, SegmentManager.scala:81-87 (read-only mmap)
        if os.path.getsize(path) == 0:
            return np.zeros(0, dtype=np.uint8)
        return np.memmap(path, dtype=np.uint8, mode="r")

    def getTable(self, tableName: str) -> Table:
        for t in self.tables:
            if t.name == tableName:
                return t
        raise Exception(f"Table {tableName} does not exist in SegmentManager")

    def getTableSegmentCount(self, tableName: str) -> int:
        t = self.getTable(tableName)
        return len(self.segments[f"{tableName}.{t.columns[0].name}"])

    def getSegment(self, id: int, tableName: str, columnName: str) -> Segment:
        key = f"{tableName}.{columnName}"
        return Segment(id, self.segments[key][id], self.segmentsMeta[key][id])

    def getSegments(self, tableName: str, columnName: str) -> List[Segment]:
        key = f"{tableName}.{columnName}"
        return [self.getSegment(i, tableName, columnName) for i in range(len(self.segments[key]))]
