"""ctypes binding of the C CPU oracle (oracle/imm3_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package (immutable3_amd/).  Parity statement: see
oracle/imm3_oracle.h (unpinned by reference tests -- the reference has none; pinned by SURVEY
Appendix B known-answer vectors and the independent numpy restatement oracle_np.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libimm3_oracle.so")

PFOR_INT, DENSE_INT, DENSE_TINYINT, DENSE_STRING = 0, 1, 2, 3
MATCH, NOTMATCH, EQ, GT, LT, NOOP = 0, 1, 2, 3, 4, 5

OK, ERR_UNSUPPORTED_CONDITION, ERR_UNSUPPORTED_VECTOR, ERR_NO_CODEC, ERR_INDEX, ERR_ARG = range(6)


class OracleError(Exception):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[{code}] {msg}")
        self.code = code
        self.msg = msg


class _CColumn(C.Structure):
    _fields_ = [
        ("dat", C.c_void_p),
        ("dat_bytes", C.c_uint64),
        ("block_offsets", C.c_void_p),
        ("n_offsets", C.c_int32),
        ("codec", C.c_int32),
        ("width", C.c_int32),
    ]


class _CSelect(C.Structure):
    _fields_ = [
        ("column", C.c_int32),
        ("cond", C.c_int32),
        ("value", C.c_double),
        ("match_bytes", C.c_void_p),
        ("match_lens", C.c_void_p),
        ("n_match", C.c_int32),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with its own Makefile (gcc -O2).  Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in ("imm3_oracle.c", "imm3_oracle_pfor.c", "imm3_oracle_snappy.c", "imm3_oracle.h")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.imm3o_bytes_to_int.restype = C.c_int32
        L.imm3o_bytes_to_int.argtypes = [C.c_void_p]
        L.imm3o_int_to_bytes.restype = None
        L.imm3o_int_to_bytes.argtypes = [C.c_int32, C.c_void_p]
        L.imm3o_d2i.restype = C.c_int32
        L.imm3o_d2i.argtypes = [C.c_double]
        L.imm3o_d2b.restype = C.c_int8
        L.imm3o_d2b.argtypes = [C.c_double]
        L.imm3o_n_batches.restype = C.c_int32
        L.imm3o_n_batches.argtypes = [C.POINTER(_CColumn)]
        L.imm3o_layout.restype = C.c_int64
        L.imm3o_layout.argtypes = [C.POINTER(_CColumn), C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.imm3o_scan_select.restype = C.c_int
        L.imm3o_scan_select.argtypes = [
            C.POINTER(_CColumn), C.c_int32, C.POINTER(_CSelect), C.c_int32, C.c_int32, C.c_int32,
            C.c_void_p, C.POINTER(C.c_uint64), C.c_char_p,
        ]
        L.imm3o_project.restype = C.c_int64
        L.imm3o_project.argtypes = [
            C.POINTER(_CColumn), C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p,
            C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int64, C.POINTER(C.c_int32),
        ]
        L.imm3o_project_agg.restype = C.c_int64
        L.imm3o_project_agg.argtypes = [C.POINTER(_CColumn), C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_char_p]
        L.imm3o_pfor_encode_bound.restype = C.c_int64
        L.imm3o_pfor_encode_bound.argtypes = [C.c_int32]
        L.imm3o_pfor_encode_block.restype = C.c_int64
        L.imm3o_pfor_encode_block.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        L.imm3o_pfor_block_count.restype = C.c_int32
        L.imm3o_pfor_block_count.argtypes = [C.c_void_p, C.c_int64]
        L.imm3o_pfor_decode_block.restype = C.c_int32
        L.imm3o_pfor_decode_block.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]
        for name in ("imm3o_crc32c", "imm3o_crc32c_masked"):
            getattr(L, name).restype = C.c_uint32
            getattr(L, name).argtypes = [C.c_void_p, C.c_int64]
        for name in ("imm3o_snappy_raw_bound", "imm3o_snappy_block_bound"):
            getattr(L, name).restype = C.c_int64
            getattr(L, name).argtypes = [C.c_int64]
        for name in ("imm3o_snappy_raw_encode", "imm3o_snappy_raw_decode", "imm3o_snappy_block_encode", "imm3o_snappy_block_decode"):
            getattr(L, name).restype = C.c_int64
            getattr(L, name).argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        for name in ("imm3o_snappy_raw_uncompressed_length", "imm3o_snappy_block_length"):
            getattr(L, name).restype = C.c_int64
            getattr(L, name).argtypes = [C.c_void_p, C.c_int64]
        _lib = L
    return _lib


@dataclass
class OColumn:
    """One used column of one segment: raw .dat bytes + .meta blockOffset table."""
    dat: np.ndarray            # uint8, C-contiguous
    block_offsets: np.ndarray  # int32, N+1 entries
    codec: int
    width: int

    def __post_init__(self):
        self.dat = np.ascontiguousarray(np.frombuffer(self.dat, dtype=np.uint8) if not isinstance(self.dat, np.ndarray) else self.dat.view(np.uint8).reshape(-1))
        self.block_offsets = np.ascontiguousarray(self.block_offsets, dtype=np.int32)


# a select leaf: (column index, cond, operand) where operand is a float (GT/LT/EQ) or a list of bytes (MATCH)
SelectSpec = Tuple[int, int, Union[float, Sequence[bytes], None]]


def bytes_to_int(b: bytes) -> int:
    buf = (C.c_uint8 * 4)(*b)
    return lib().imm3o_bytes_to_int(buf)


def int_to_bytes(v: int) -> bytes:
    buf = (C.c_uint8 * 4)()
    lib().imm3o_int_to_bytes(v, buf)
    return bytes(buf)


def d2i(d: float) -> int:
    return lib().imm3o_d2i(d)


def d2b(d: float) -> int:
    return lib().imm3o_d2b(d)


def _ccols(cols: Sequence[OColumn]):
    arr = (_CColumn * len(cols))()
    for i, c in enumerate(cols):
        arr[i].dat = c.dat.ctypes.data
        arr[i].dat_bytes = c.dat.size
        arr[i].block_offsets = c.block_offsets.ctypes.data
        arr[i].n_offsets = c.block_offsets.size
        arr[i].codec = c.codec
        arr[i].width = c.width
    return arr


def _csels(sels: Sequence[SelectSpec]):
    arr = (_CSelect * max(1, len(sels)))()
    keep = []
    for i, (col, cond, operand) in enumerate(sels):
        arr[i].column = col
        arr[i].cond = cond
        arr[i].value = 0.0
        arr[i].n_match = 0
        if cond in (MATCH, NOTMATCH):
            vals = [bytes(v) for v in (operand or [])]
            blob = np.frombuffer(b"".join(vals) or b"\0", dtype=np.uint8).copy()
            lens = np.array([len(v) for v in vals] or [0], dtype=np.int32)
            keep += [blob, lens]
            arr[i].match_bytes = blob.ctypes.data
            arr[i].match_lens = lens.ctypes.data
            arr[i].n_match = len(vals)
        elif operand is not None:
            arr[i].value = float(operand)
    return arr, keep


def layout(first: OColumn, block_size: int):
    """(batch_size[int32], batch_oid[int32], batch_word_off[int64], total_words)"""
    cc = _ccols([first])
    nb = lib().imm3o_n_batches(cc)
    size = np.zeros(max(nb, 1), dtype=np.int32)
    oid = np.zeros(max(nb, 1), dtype=np.int32)
    woff = np.zeros(max(nb, 1), dtype=np.int64)
    tw = lib().imm3o_layout(cc, block_size, size.ctypes.data, oid.ctypes.data, woff.ctypes.data)
    return size[:nb], oid[:nb], woff[:nb], int(tw)


def scan_select(cols: Sequence[OColumn], sels: Sequence[SelectSpec], block_size: int, flavour: int = 1):
    """ScanOp -> SelectOp* over one segment.  Returns (words[uint64, batch-major], count)."""
    _, _, _, tw = layout(cols[0], block_size)
    words = np.zeros(max(tw, 1), dtype=np.uint64)
    count = C.c_uint64(0)
    msg = C.create_string_buffer(256)
    cc = _ccols(cols)
    cs, keep = _csels(sels)
    rc = lib().imm3o_scan_select(cc, len(cols), cs, len(sels), block_size, flavour,
                                 words.ctypes.data, C.byref(count), msg)
    del keep
    if rc != OK:
        raise OracleError(rc, msg.value.decode())
    return words[:tw], int(count.value)


def project(cols: Sequence[OColumn], proj: Sequence[int], limit: int, block_size: int, words: np.ndarray,
            cap_rows: Optional[int] = None):
    """ProjectOp.  Returns (n_rows, batch[int32], pos[int32], [per-proj-col uint8 array (n,width)], would_throw)."""
    words = np.ascontiguousarray(words, dtype=np.uint64)
    if cap_rows is None:
        cap_rows = int(sum(bin(int(w)).count("1") for w in words)) if words.size < 4096 else int(
            np.unpackbits(words.view(np.uint8)).sum())
        if limit > 0:
            cap_rows = min(cap_rows, limit)
    cap = max(cap_rows, 1)
    batch = np.zeros(cap, dtype=np.int32)
    pos = np.zeros(cap, dtype=np.int32)
    widths = [4 if cols[j].codec == DENSE_INT else cols[j].width for j in proj]
    vals = [np.zeros((cap, w), dtype=np.uint8) for w in widths]
    ptrs = (C.c_void_p * max(1, len(proj)))(*[v.ctypes.data for v in vals])
    projarr = np.array(list(proj) or [0], dtype=np.int32)
    wt = C.c_int32(0)
    cc = _ccols(cols)
    n = lib().imm3o_project(cc, len(cols), projarr.ctypes.data, len(proj), limit, block_size,
                            words.ctypes.data if words.size else None, batch.ctypes.data, pos.ctypes.data,
                            ptrs, cap_rows, C.byref(wt))
    if n < 0:
        raise OracleError(int(-n), "project failed")
    n = int(n)
    return n, batch[:n], pos[:n], [v[:n] for v in vals], bool(wt.value)


# ---- PFOR_INT block codec (imm3_oracle_pfor.c) ----------------------------------------------------------------
AGG_KIND = {"count": 0, "min": 1, "max": 2}
DOUBLE_MAX = 1.7976931348623157e308


def project_agg(cols: Sequence[OColumn], group: Sequence[int], aggs, words: np.ndarray, max_groups: int = 1 << 20):
    """ProjectAggOp over one segment (the C twin of oracle_np.project_agg).  aggs: [(kind in {'count','min','max'}, column)].
    Returns the same insertion-ordered dict: groupKey -> [int (count) | float (min / max) | str (MaxStringAggr)]."""
    words = np.ascontiguousarray(words, dtype=np.uint64)
    na = len(aggs)
    key_stride, str_stride = 96, 40
    garr = np.array(list(group) or [0], dtype=np.int32)
    aarr = np.array([[AGG_KIND[k], c] for k, c in aggs], dtype=np.int32)
    cap = max_groups
    while True:
        keys = np.zeros(cap * key_stride, dtype=np.uint8)
        counts = np.zeros(cap * na, dtype=np.int64)
        nums = np.zeros(cap * na, dtype=np.float64)
        strs = np.zeros(cap * na * str_stride, dtype=np.uint8)
        msg = C.create_string_buffer(256)
        cc = _ccols(cols)
        n = lib().imm3o_project_agg(cc, len(cols), garr.ctypes.data, len(group), aarr.ctypes.data, na,
                                    words.ctypes.data if words.size else None, keys.ctypes.data, key_stride, counts.ctypes.data,
                                    nums.ctypes.data, strs.ctypes.data, str_stride, cap, msg)
        if n < 0:
            raise OracleError(int(-n), msg.value.decode() or "project_agg failed")
        break
    out = {}
    for g in range(int(n)):
        kb = keys[g * key_stride:(g + 1) * key_stride].tobytes()
        key = kb[:kb.index(b"\0")].decode("utf-8", errors="replace")
        st = []
        for a, (kind, c) in enumerate(aggs):
            if kind == "count":
                st.append(int(counts[g * na + a]))
            elif cols[c].codec == DENSE_STRING:
                sb = strs[(g * na + a) * str_stride:(g * na + a + 1) * str_stride].tobytes()
                st.append(sb[:sb.index(b"\0")].decode("utf-8", errors="replace"))
            else:
                st.append(float(nums[g * na + a]))
        out[key] = st
    return out


def pfor_encode_block(vals: np.ndarray) -> bytes:
    """PFORCodecInt.encode of one block of int32 values (PFORCodec.scala:19-31)."""
    v = np.ascontiguousarray(vals, dtype=np.int32)
    cap = lib().imm3o_pfor_encode_bound(v.size)
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().imm3o_pfor_encode_block(v.ctypes.data, v.size, out.ctypes.data, cap)
    if n < 0:
        raise OracleError(ERR_ARG, "pfor encode failed")
    return out[:n].tobytes()


def pfor_decode_block(blk: bytes) -> np.ndarray:
    """The decode the encoder implies (IntegratedIntCompressor.uncompress of the block's big-endian words)."""
    b = np.frombuffer(blk, dtype=np.uint8)
    n = lib().imm3o_pfor_block_count(b.ctypes.data, b.size)
    if n < 0:
        raise OracleError(ERR_INDEX, "pfor block too short")
    out = np.zeros(max(n, 1), dtype=np.int32)
    got = lib().imm3o_pfor_decode_block(b.ctypes.data, b.size, out.ctypes.data, out.size)
    if got < 0:
        raise OracleError(ERR_INDEX, "malformed pfor block")
    return out[:got]


def pfor_encode_column(vals: np.ndarray, block_rows: int):
    """Whole column -> (.dat bytes, blockOffset table), one encoded block per block_rows values (SegmentWriter.flush)."""
    v = np.ascontiguousarray(vals, dtype=np.int32)
    parts, offs = [], [0]
    for s in range(0, v.size, block_rows):
        parts.append(pfor_encode_block(v[s:s + block_rows]))
        offs.append(offs[-1] + len(parts[-1]))
    return np.frombuffer(b"".join(parts), dtype=np.uint8).copy(), np.array(offs, dtype=np.int32)


def pfor_decode_column(dat: np.ndarray, offs: np.ndarray) -> np.ndarray:
    d = np.ascontiguousarray(dat, dtype=np.uint8)
    out = [pfor_decode_block(d[int(offs[k]):int(offs[k + 1])].tobytes()) for k in range(len(offs) - 1)]
    return np.concatenate(out) if out else np.zeros(0, dtype=np.int32)


# ---- snappy-coded blocks (imm3_oracle_snappy.c) ------------------------------------------------------------
def _buf(b) -> np.ndarray:
    return np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else np.ascontiguousarray(b, dtype=np.uint8)


def crc32c(b) -> int:
    a = _buf(b)
    return int(lib().imm3o_crc32c(a.ctypes.data, a.size))


def crc32c_masked(b) -> int:
    a = _buf(b)
    return int(lib().imm3o_crc32c_masked(a.ctypes.data, a.size))


def _codec_call(fn, bound, b) -> bytes:
    a = _buf(b)
    out = np.zeros(max(int(bound), 1), dtype=np.uint8)
    n = fn(a.ctypes.data, a.size, out.ctypes.data, out.size)
    if n < 0:
        raise OracleError(ERR_INDEX, "malformed snappy data")
    return out[:n].tobytes()


def snappy_raw_encode(b) -> bytes:
    return _codec_call(lib().imm3o_snappy_raw_encode, lib().imm3o_snappy_raw_bound(len(_buf(b))), b)


def snappy_raw_decode(b) -> bytes:
    a = _buf(b)
    n = lib().imm3o_snappy_raw_uncompressed_length(a.ctypes.data, a.size)
    if n < 0:
        raise OracleError(ERR_INDEX, "malformed snappy preamble")
    return _codec_call(lib().imm3o_snappy_raw_decode, n, b)


def snappy_block_encode(b) -> bytes:
    """SnappyCodec.encode of one block's raw value bytes (SnappyCodec.scala:15-27): SnappyOutputStream framing."""
    return _codec_call(lib().imm3o_snappy_block_encode, lib().imm3o_snappy_block_bound(len(_buf(b))), b)


def snappy_block_decode(b) -> bytes:
    a = _buf(b)
    n = lib().imm3o_snappy_block_length(a.ctypes.data, a.size)
    if n < 0:
        raise OracleError(ERR_INDEX, "malformed snappy block")
    return _codec_call(lib().imm3o_snappy_block_decode, n, b)
