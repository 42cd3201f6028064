"""Second, independent CPU restatement of the reference hot path in numpy / pure Python.

TEST INFRASTRUCTURE ONLY (same rule as oracle_c.py: tests/, smoke() and bench's cpu_baseline leg).
Written from the reference's Scala semantics without looking at imm3_oracle.c's structure: it
works on whole decoded vectors with numpy instead of per-row loops, so an error shared by both
restatements would have to be an error in reading the reference, not a coding slip.  Parity vs
the reference itself is UNPINNED by reference tests (the reference has none; see imm3_oracle.h).

Citations: core/ = core/src/main/scala/immutabledb, engine/ = engine/src/main/scala/immutabledb.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np

DENSE_INT, DENSE_TINYINT, DENSE_STRING = 1, 2, 3
MATCH, NOTMATCH, EQ, GT, LT, NOOP = 0, 1, 2, 3, 4, 5


class RefException(Exception):
    """Stands for the reference's `throw new Exception(msg)` / JVM runtime exceptions."""


def to_int(d: float) -> int:
    """Scala Double.toInt (JVM d2i): NaN->0, saturating, truncation toward zero (Select.scala:65)."""
    if math.isnan(d):
        return 0
    if d >= 2147483647:
        return 2147483647
    if d <= -2147483648:
        return -2147483648
    return int(math.trunc(d))


def to_byte(d: float) -> int:
    """Scala Double.toByte: toInt then keep the low 8 bits as a signed byte (Select.scala:73)."""
    b = to_int(d) & 0xFF
    return b - 256 if b >= 128 else b


def bytes_to_int(b: bytes) -> int:
    """Conversions.bytesToInt (core/util/Conversions.scala:17-24): ((((b3)<<8 + b2)<<8 + b1)<<8) + b0, 32-bit wrap."""
    r = 0
    for i in (3, 2, 1):
        r = ((r + (b[i] & 0xFF)) << 8) & 0xFFFFFFFF
    r = (r + (b[0] & 0xFF)) & 0xFFFFFFFF
    return r - (1 << 32) if r & 0x80000000 else r


def int_to_bytes(v: int) -> bytes:
    """IntType.valueToBytes (core/DataType.scala:40-47)."""
    v &= 0xFFFFFFFF
    return bytes([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF, (v >> 24) & 0xFF])


def _block_bounds(offsets: np.ndarray) -> List[Tuple[int, int]]:
    """Segment.BlockIterator (core/storage/Segment.scala:158-180): block k holds
    blockOffsets(k+1)-blockOffsets(k) bytes taken by RELATIVE gets from a rewound buffer."""
    lens = np.diff(offsets.astype(np.int64))
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]]) if lens.size else np.zeros(0, dtype=np.int64)
    return [(int(s), int(l)) for s, l in zip(starts, lens)]


def decode_block(raw: np.ndarray, codec: int, width: int) -> np.ndarray:
    """DenseCodec*.decode (core/codec/DenseCodec.scala:37-73).  Returns int32[n], int8[n] or uint8[n,width].
    A trailing partial element re-uses the previous chunk's tail bytes (read() count ignored)."""
    n_full = raw.size // width
    rem = raw.size - n_full * width
    body = raw[: n_full * width]
    if rem:
        prev = raw[(n_full - 1) * width: n_full * width] if n_full else np.zeros(width, dtype=np.uint8)
        last = prev.copy()
        last[:rem] = raw[n_full * width:]
        body = np.concatenate([body, last])
    if codec == DENSE_INT:
        return body.view("<i4").astype(np.int32)
    if codec == DENSE_TINYINT:
        return body.view(np.int8)
    if codec == DENSE_STRING:
        return body.reshape(-1, width)
    raise RefException(f"No implementation for {codec}")  # Scan.scala:49


def layout(first_offsets: np.ndarray, first_width: int, block_size: int):
    sizes, oids, woffs = [], [], []
    w = 0
    for k, (_, ln) in enumerate(_block_bounds(first_offsets)):
        n = -(-ln // first_width) if ln > 0 else 0
        sizes.append(n)
        oids.append(k * block_size)  # vecCounter * table.blockSize, Scan.scala:60
        woffs.append(w)
        w += -(-n // 64)
    return np.array(sizes, np.int32), np.array(oids, np.int32), np.array(woffs, np.int64), w


def _predicate(vec: np.ndarray, codec: int, width: int, cond: int, operand) -> np.ndarray:
    """Boolean KEEP mask of one SelectOp over one decoded vector (Select.scala:25-165)."""
    if cond in (GT, LT, EQ):
        if codec == DENSE_INT:
            t = np.int32(to_int(float(operand)))
        elif codec == DENSE_TINYINT:
            t = np.int8(to_byte(float(operand)))
        else:
            raise RefException("Unsupported column vector")
        if cond == GT:
            return vec > t
        if cond == LT:
            return vec < t
        return vec == t
    if cond == MATCH:
        if codec != DENSE_STRING:
            raise RefException("Unsupported column vector")
        keep = np.zeros(vec.shape[0], dtype=bool)
        for v in operand:
            v = bytes(v)
            if len(v) == width:
                keep |= (vec == np.frombuffer(v, dtype=np.uint8)).all(axis=1)
        return keep
    raise RefException(f"Unsupported condition: {cond}")  # Select.scala:22


def scan_select(cols: Sequence[Tuple[np.ndarray, np.ndarray, int, int]], sels, block_size: int):
    """cols: [(dat uint8, offsets int32, codec, width)] in used-column order; sels: [(col, cond, operand)].
    Returns (words uint64 batch-major, count, per-batch list of bool keep masks)."""
    for (_, cond, _) in sels:  # SelectOp.iterator rejects these when the chain is built (Select.scala:17-23)
        if cond not in (MATCH, GT, LT, EQ):
            raise RefException(f"Unsupported condition: {cond}")
    bounds = [_block_bounds(c[1]) for c in cols]
    nb = len(bounds[0])
    words: List[np.ndarray] = []
    masks: List[np.ndarray] = []
    count = 0
    for k in range(nb):
        vecs = []
        for ci, (dat, _, codec, width) in enumerate(cols):
            if k >= len(bounds[ci]):
                raise RefException("ArrayIndexOutOfBounds")
            s, ln = bounds[ci][k]
            if codec not in (DENSE_INT, DENSE_TINYINT, DENSE_STRING):
                raise RefException(f"No implementation for {codec}")  # Scan.scala:49, first batch only
            if ln < 0 or s + ln > dat.size:
                raise RefException("BufferUnderflow")
            vecs.append(decode_block(dat[s: s + ln], codec, width))
        size = vecs[0].shape[0]  # Scan.scala:55
        keep = np.ones(size, dtype=bool)  # Scan.scala:56-57
        for (ci, cond, operand) in sels:
            if cond in (NOTMATCH, NOOP):
                raise RefException(f"Unsupported condition: {cond}")
            v = vecs[ci]
            codec, width = cols[ci][2], cols[ci][3]
            if cond in (GT, LT, EQ) and codec == DENSE_STRING or cond == MATCH and codec != DENSE_STRING:
                raise RefException("Unsupported column vector")
            if v.shape[0] < size:
                raise RefException("ArrayIndexOutOfBounds")
            keep &= _predicate(v[:size], codec, width, cond, operand)
        masks.append(keep)
        count += int(keep.sum())
        nw = -(-size // 64)
        padded = np.zeros(nw * 64, dtype=np.uint8)
        padded[:size] = keep
        # mutable.BitSet: bit i -> word i>>6, bit i&63  == little-endian bit packing
        words.append(np.packbits(padded, bitorder="little").view("<u8") if nw else np.zeros(0, np.uint64))
    allw = np.concatenate(words).astype(np.uint64) if words else np.zeros(0, np.uint64)
    return allw, count, masks


def project(cols, proj: Sequence[int], limit: int, masks: Sequence[np.ndarray]):
    """ProjectOp (engine/engine/operator/Project.scala:37-80) with empty batches skipped.
    Returns (rows: list of tuples in SELECT-list order, [(batch,pos)], would_throw)."""
    bounds = [_block_bounds(c[1]) for c in cols]
    rows, where = [], []
    would_throw = False
    for k, keep in enumerate(masks):
        if limit > 0 and len(rows) >= limit:
            break
        idx = np.flatnonzero(keep)
        if idx.size == 0:
            would_throw = True
            continue
        vecs = {}
        for j in proj:
            dat, _, codec, width = cols[j]
            s, ln = bounds[j][k]
            vecs[j] = decode_block(dat[s: s + ln], codec, width)
        for p in idx:
            if limit > 0 and len(rows) >= limit:
                break
            r = []
            for j in proj:
                v = vecs[j]
                if p >= v.shape[0]:
                    raise RefException("ArrayIndexOutOfBounds")
                r.append(bytes(v[p]) if cols[j][2] == DENSE_STRING else int(v[p]))
            rows.append(tuple(r))
            where.append((k, int(p)))
    return rows, where, would_throw


# ------------------------------------------------------------------------------------------------------------
# ProjectAggOp (engine/src/main/scala/immutabledb/engine/operator/ProjectAggregate.scala:115-227) and the
# cross-segment combine of ProjectAggregateQueueOp (engine/.../operator/ProjectAggregateQueue.scala:9-55).
# ------------------------------------------------------------------------------------------------------------
def java_double_to_string(v: float) -> str:
    """Double.toString for the integral values the aggregators hold (value.toDouble of an Int / Byte)."""
    if v != v:
        return "NaN"
    iv = int(v)
    if abs(iv) < 10 ** 7:
        return f"{iv}.0" if iv != 0 or math.copysign(1.0, v) > 0 else "-0.0"
    digits = str(abs(iv))
    mant = digits[0] + "." + (digits[1:].rstrip("0") or "0")
    return ("-" if iv < 0 else "") + mant + "E" + str(len(digits) - 1)


DOUBLE_MIN_VALUE = 4.9e-324              # scala Double.MinValue is -Double.MaxValue; see below
DOUBLE_MAX = 1.7976931348623157e308


def project_agg(cols, group: Sequence[int], aggs: Sequence[Tuple[str, int]], masks: Sequence[np.ndarray]):
    """One segment.  cols: used columns (dat, offsets, codec, width) in batch order; group: used indices of the
    group-by columns IN BATCH-COLUMN ORDER (groupCols is built by filtering currVecBatch.columns, :135-140);
    aggs: [(kind in {'count','min','max'}, used index)] in SELECT-list order; masks: per-batch keep masks.
    Returns an insertion-ordered dict  groupKey -> [state per aggregate]  (LinkedHashMap, :126) where a state is
    an int (CountAggr), a float (Max/MinDoubleAggr, started at -/+Double.MaxValue) or a str (MaxStringAggr)."""
    bounds = [_block_bounds(c[1]) for c in cols]
    result = {}
    for k, keep in enumerate(masks):
        idx = np.flatnonzero(keep)
        if idx.size == 0:
            continue
        need = sorted(set(group) | {c for _, c in aggs})
        vecs = {}
        for j in need:
            dat, _, codec, width = cols[j]
            s, ln = bounds[j][k]
            vecs[j] = decode_block(dat[s: s + ln], codec, width)

        def val(j, p):
            v = vecs[j][p]
            return bytes(v).decode("utf-8", errors="replace") if cols[j][2] == DENSE_STRING else int(v)

        for p in idx:
            key = "_".join(str(val(j, p)) for j in group)          # getResultMapKey: mkString("_"), :144
            st = result.get(key)
            if st is None:
                st = []
                for kind, j in aggs:                                # getNewAggs: fresh aggregators, :146-150
                    is_str = cols[j][2] == DENSE_STRING
                    if kind == "count":
                        st.append(0)
                    elif is_str:
                        st.append("")                              # MaxStringAggr.max = "", :80 (also what Min maps to, Engine.scala:145)
                    elif kind == "max":
                        st.append(-DOUBLE_MAX)                     # Double.MinValue, :38
                    else:
                        st.append(DOUBLE_MAX)                      # Double.MaxValue, :51
                result[key] = st
            for a, (kind, j) in enumerate(aggs):
                v = val(j, p)
                if kind == "count":
                    st[a] += 1                                     # CountAggr.add, :24
                elif isinstance(v, str):
                    if kind == "min":
                        raise RefException("bad aggregator for this data type")
                    st[a] = v if st[a] == "" or v > st[a] else st[a]   # MaxStringAggr.add, :81-83
                elif kind == "max":
                    st[a] = float(v) if float(v) > st[a] else st[a]    # MaxDoubleAggr.add, :39
                else:
                    st[a] = float(v) if float(v) < st[a] else st[a]    # MinDoubleAggr.add, :52
    return result


def combine_agg(per_segment, aggs: Sequence[Tuple[str, int]]):
    """ProjectAggregateQueueOp.init: merge the per-segment maps by key, first arrival first (segments in
    ascending order here; the reference's arrival order is a race).  Returns ordered dict key -> states."""
    out = {}
    for seg in per_segment:
        for key, st in seg.items():
            cur = out.get(key)
            if cur is None:
                out[key] = list(st)
                continue
            for a, (kind, _) in enumerate(aggs):
                if kind == "count":
                    cur[a] += st[a]
                elif isinstance(st[a], str):
                    cur[a] = st[a] if cur[a] == "" or st[a] > cur[a] else cur[a]
                elif kind == "max":
                    cur[a] = max(cur[a], st[a])
                else:
                    cur[a] = min(cur[a], st[a])
    return out


def agg_repr(state) -> str:
    """Aggregator.repr: Long.toString / Double.toString / the String."""
    if isinstance(state, float):
        return java_double_to_string(state)
    return str(state)


# ------------------------------------------------------------------------------------------
# PFOR_INT block codec -- second, independent restatement (bit-stream formulation; the C oracle works word by word).
# Format: core/codec/PFORCodec.scala:19-31 over JavaFastPFOR 0.1.10's IntegratedIntCompressor (see
# imm3_oracle_pfor.c for the statement of the library's algorithm).  TEST INFRASTRUCTURE ONLY; parity unpinned.
# ------------------------------------------------------------------------------------------
def _pfor_width(deltas_u32: np.ndarray) -> int:
    m = int(np.bitwise_or.reduce(deltas_u32.astype(np.uint64)))
    return m.bit_length()


def _pfor_pack(values_u32: np.ndarray, init: int, b: int) -> np.ndarray:
    """32 values -> b little-endian words (b == 32: the values themselves)."""
    if b == 32:
        return values_u32.astype(np.uint32)
    if b == 0:
        return np.zeros(0, dtype=np.uint32)
    prev = np.concatenate([[np.uint32(init & 0xFFFFFFFF)], values_u32[:-1]]).astype(np.uint32)
    d = (values_u32 - prev).astype(np.uint32)  # wrapping
    bits = ((d[:, None] >> np.arange(b, dtype=np.uint32)[None, :]) & 1).astype(np.uint8).reshape(-1)  # LSB-first stream
    return np.packbits(bits, bitorder="little").view("<u4").astype(np.uint32)


def pfor_encode_block(vals: np.ndarray) -> bytes:
    v = np.ascontiguousarray(vals, dtype=np.int32).view(np.uint32)
    n = v.size
    words = [np.array([n], dtype=np.uint32)]
    init = 0
    n_mini = n // 32
    widths, packed = [], []
    for m in range(n_mini):
        blk = v[32 * m:32 * m + 32]
        prev = np.concatenate([[np.uint32(init)], blk[:-1]]).astype(np.uint32)
        b = _pfor_width((blk - prev).astype(np.uint32))
        widths.append(b)
        packed.append(_pfor_pack(blk, init, b))
        init = int(blk[-1])
    g = 0
    while g + 4 <= n_mini:
        words.append(np.array([(widths[g] << 24) | (widths[g + 1] << 16) | (widths[g + 2] << 8) | widths[g + 3]], dtype=np.uint32))
        words += packed[g:g + 4]
        g += 4
    for m in range(g, n_mini):
        words.append(np.array([widths[m]], dtype=np.uint32))
        words.append(packed[m])
    tail = v[32 * n_mini:]
    if tail.size:
        out = bytearray()
        for x in tail:
            d = (int(x) - init) & 0xFFFFFFFF
            init = int(x)
            while d >= 128:
                out.append(d & 127)
                d >>= 7
            out.append(d | 128)
        while len(out) % 4:
            out.append(0)
        words.append(np.frombuffer(bytes(out), dtype="<u4").astype(np.uint32))
    w = np.concatenate(words).astype(">u4")  # ByteBuffer.putInt: big-endian
    return w.tobytes() + b"\0" * 8           # result.array(): the 8 spare bytes of allocate(len * 4 + 8)


def pfor_decode_block(blk: bytes) -> np.ndarray:
    w = np.frombuffer(blk, dtype=">u4").astype(np.uint32)
    n = int(w[0])
    out = np.zeros(n, dtype=np.uint32)
    pos, init, n_mini = 1, 0, n // 32

    def unpack(pos, b, init):
        if b == 32:
            return w[pos:pos + 32].copy()
        if b == 0:
            return np.full(32, init, dtype=np.uint32)
        bits = np.unpackbits(w[pos:pos + b].astype("<u4").view(np.uint8), bitorder="little").reshape(32, b)
        d = (bits.astype(np.uint64) << np.arange(b, dtype=np.uint64)[None, :]).sum(axis=1)
        return ((np.cumsum(d) + init) & 0xFFFFFFFF).astype(np.uint32)

    m = 0
    while m + 4 <= n_mini:
        h = int(w[pos]); pos += 1
        for k in range(4):
            b = (h >> (24 - 8 * k)) & 255
            out[32 * (m + k):32 * (m + k) + 32] = unpack(pos, b, init)
            pos += b
            init = int(out[32 * (m + k) + 31])
        m += 4
    while m < n_mini:
        b = int(w[pos]); pos += 1
        out[32 * m:32 * m + 32] = unpack(pos, b, init)
        pos += b
        init = int(out[32 * m + 31])
        m += 1
    if n > 32 * n_mini:
        by = w[pos:].astype("<u4").view(np.uint8)
        i = 0
        for k in range(32 * n_mini, n):
            val, shift = 0, 0
            while True:
                c = int(by[i]); i += 1
                val += (c & 127) << shift
                if c & 128:
                    break
                shift += 7
            init = (init + val) & 0xFFFFFFFF
            out[k] = init
    return out.view(np.int32)


# ------------------------------------------------------------------------------------------
# snappy-coded blocks -- second, independent restatement (table-driven CRC, slice-based decoder).  Format: what
# SnappyCodec.encode writes (core/codec/SnappyCodec.scala:14-43) = iq80 snappy 0.4 SnappyOutputStream framing around
# raw Snappy; see imm3_oracle_snappy.c for the statement of the format.  TEST INFRASTRUCTURE ONLY; parity unpinned.
# ------------------------------------------------------------------------------------------
_CRC32C_TABLE = None


def _crc32c_table():
    global _CRC32C_TABLE
    if _CRC32C_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t.append(c)
        _CRC32C_TABLE = t
    return _CRC32C_TABLE


def crc32c(data: bytes) -> int:
    t = _crc32c_table()
    c = 0xFFFFFFFF
    for b in data:
        c = t[(c ^ b) & 255] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def crc32c_masked(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def snappy_raw_decode(data: bytes) -> bytes:
    ip, want, shift = 0, 0, 0
    while True:
        b = data[ip]; ip += 1
        want |= (b & 127) << shift
        if not b & 128:
            break
        shift += 7
    out = bytearray()
    while ip < len(data):
        tag = data[ip]; ip += 1
        kind = tag & 3
        if kind == 0:
            ln = (tag >> 2) + 1
            if ln > 60:
                nb = ln - 60
                ln = int.from_bytes(data[ip:ip + nb], "little") + 1
                ip += nb
            out += data[ip:ip + ln]
            ip += ln
            continue
        if kind == 1:
            ln, off = 4 + ((tag >> 2) & 7), ((tag >> 5) << 8) | data[ip]
            ip += 1
        elif kind == 2:
            ln, off = 1 + (tag >> 2), int.from_bytes(data[ip:ip + 2], "little")
            ip += 2
        else:
            ln, off = 1 + (tag >> 2), int.from_bytes(data[ip:ip + 4], "little")
            ip += 4
        assert 0 < off <= len(out)
        start = len(out) - off
        pattern = bytes(out[start:start + min(off, ln)])       # an overlapping copy repeats its first `off` bytes
        out += (pattern * (ln // len(pattern) + 1))[:ln]
    assert len(out) == want
    return bytes(out)


def snappy_block_decode(blk: bytes) -> bytes:
    assert blk[:7] == b"snappy\x00"
    ip, out = 7, bytearray()
    while ip < len(blk):
        flag, plen, crc = blk[ip], int.from_bytes(blk[ip + 1:ip + 3], "big"), int.from_bytes(blk[ip + 3:ip + 7], "big")
        ip += 7
        payload = blk[ip:ip + plen]
        assert len(payload) == plen and flag in (0, 1)
        chunk = snappy_raw_decode(payload) if flag else bytes(payload)
        assert crc32c_masked(chunk) == crc
        out += chunk
        ip += plen
    return bytes(out)
