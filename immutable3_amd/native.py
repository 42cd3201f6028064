"""ctypes binding of libimm3.so (include/imm3.h) -- the only way the host side reaches the GPU.

There is no CPU fallback: if the shared library is missing, or no HIP device is present, the
calls raise.  The oracle under oracle/ is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref
from typing import List, Optional, Sequence

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libimm3.so")

# CodecType (core/src/main/scala/immutabledb/codec/Codec.scala:21-24)
PFOR_INT, DENSE_INT, DENSE_TINYINT, DENSE_STRING = 0, 1, 2, 3
SNAPPY_INT, SNAPPY_TINYINT, SNAPPY_STRING = 16, 17, 18   # extension: blocks as SnappyCodec.encode writes them
INT_CODECS, TINYINT_CODECS = (DENSE_INT, PFOR_INT, SNAPPY_INT), (DENSE_TINYINT, SNAPPY_TINYINT)
# SelectCondition (core/src/main/scala/immutabledb/Query.scala:3-9)
MATCH, NOTMATCH, EQ, GT, LT, NOOP = 0, 1, 2, 3, 4, 5

OK = 0
ERR_UNSUPPORTED_CONDITION, ERR_UNSUPPORTED_VECTOR, ERR_NO_CODEC, ERR_LAYOUT, ERR_ARG, ERR_DEVICE, ERR_STATE = 1, 2, 3, 4, 5, 6, 7

# every symbol include/imm3.h declares (tests check the library exports all of them)
EXPORTS = [
    "imm3_abi_version", "imm3_last_error", "imm3_device_count",
    "imm3_ctx_create", "imm3_ctx_destroy", "imm3_ctx_sync", "imm3_ctx_stream",
    "imm3_ctx_capture_begin", "imm3_ctx_capture_end", "imm3_graph_launch", "imm3_graph_destroy",
    "imm3_segment_create", "imm3_segment_create_async", "imm3_segment_wait", "imm3_segment_wrap_device", "imm3_segment_destroy", "imm3_segment_bytes",
    "imm3_table_create", "imm3_table_destroy", "imm3_query_create_table", "imm3_query_create_table_agg",
    "imm3_query_segment_starts", "imm3_query_locate_rows",
    "imm3_query_create", "imm3_query_create_agg", "imm3_query_group_count", "imm3_query_fetch_groups", "imm3_query_agg_shape",
    "imm3_query_destroy", "imm3_query_reserve_rows",
    "imm3_query_run", "imm3_query_run_select", "imm3_query_run_count", "imm3_query_sync", "imm3_query_join_count", "imm3_query_log_counts",
    "imm3_query_layout", "imm3_query_batches", "imm3_query_count", "imm3_query_bitmap",
    "imm3_query_row_count", "imm3_query_fetch_rows", "imm3_query_device_ptr",
    "imm3_comm_unique_id", "imm3_comm_create", "imm3_comm_create_all", "imm3_comm_destroy", "imm3_comm_info",
    "imm3_comm_sync", "imm3_comm_join", "imm3_comm_allreduce_u64", "imm3_comm_allreduce_count", "imm3_comm_allreduce_count_all", "imm3_comm_merge_groups", "imm3_comm_merge_groups_all",
    "imm3_pfor_encode_bound", "imm3_pfor_encode_block", "imm3_pfor_encode_column",
    "imm3_snappy_encode_bound", "imm3_snappy_encode_block",
]
# include/imm3_diag.h: measurement / tuning hooks, not part of the drop-in boundary
DIAG_EXPORTS = [
    "imm3_ctx_timing_enable", "imm3_ctx_timing_reset", "imm3_ctx_timing_mask", "imm3_ctx_timing_collect", "imm3_ctx_set_tuning",
    "imm3_ctx_measure_read_gbps", "imm3_ctx_devclock_enable", "imm3_ctx_devclock_collect", "imm3_ctx_devclock_raw", "imm3_query_plan",
    "imm3_ctx_inject_fault", "imm3_ctx_debug_device_lock", "imm3_plan_predict", "imm3_comm_debug_standin", "imm3_plan_limit_scan",
]
COMM_ID_BYTES = 128


class Imm3Error(Exception):
    """The reference's `throw new Exception(msg)` as surfaced by the C ABI (status + message)."""

    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code
        self.msg = msg


class CColumn(C.Structure):
    _fields_ = [
        ("codec", C.c_int32),
        ("width", C.c_int32),
        ("dat", C.c_void_p),
        ("dat_bytes", C.c_uint64),
        ("block_offsets", C.c_void_p),
        ("n_offsets", C.c_int32),
    ]


class CSelect(C.Structure):
    _fields_ = [
        ("column", C.c_int32),
        ("cond", C.c_int32),
        ("value", C.c_double),
        ("match_bytes", C.c_void_p),
        ("match_lens", C.c_void_p),
        ("n_match", C.c_int32),
    ]


_lib = None


def load() -> C.CDLL:
    """Load libimm3.so; fail loudly when the HIP extension has not been built.
    Note for hosts that also use PyTorch-ROCm in the same process: import torch BEFORE the first call of this function.
    torch bundles its own HIP runtime; the runtime loaded first serves the process, a second instance sees no GPU."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("IMM3_LIB_PATH", LIB_PATH)   # development only: A/B an older build of the SAME library on one box
    if not os.path.exists(path):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C immutable3_amd/csrc). "
            "The immutable3 GPU path has no CPU fallback."
        )
    L = C.CDLL(path)
    if path != LIB_PATH:   # an older build lacks the newer entry points: bind what is there
        class _Tolerant:
            def __init__(self, lib):
                object.__setattr__(self, "_lib", lib)

            def __getattr__(self, name):
                try:
                    return getattr(object.__getattribute__(self, "_lib"), name)
                except AttributeError:
                    class _Missing:
                        argtypes = restype = None
                    return _Missing()
        L = _Tolerant(L)
    vp, i32, i64, u64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64
    P = C.POINTER
    L.imm3_abi_version.restype = C.c_int
    L.imm3_last_error.restype = C.c_char_p
    L.imm3_device_count.argtypes = [P(C.c_int)]
    L.imm3_ctx_create.argtypes = [C.c_int, vp, P(vp)]
    L.imm3_ctx_destroy.argtypes = [vp]
    L.imm3_ctx_sync.argtypes = [vp]
    L.imm3_ctx_stream.argtypes = [vp, P(vp)]
    L.imm3_ctx_capture_begin.argtypes = [vp]
    L.imm3_ctx_capture_end.argtypes = [vp, P(vp)]
    L.imm3_graph_launch.argtypes = [vp]
    L.imm3_graph_destroy.argtypes = [vp]
    L.imm3_segment_create.argtypes = [vp, P(CColumn), i32, P(vp)]
    L.imm3_segment_wrap_device.argtypes = [vp, P(CColumn), i32, P(vp)]
    L.imm3_segment_create_async.argtypes = [vp, P(CColumn), i32, P(vp)]
    L.imm3_segment_wait.argtypes = [vp]
    L.imm3_segment_destroy.argtypes = [vp]
    L.imm3_segment_bytes.argtypes = [vp, P(u64)]
    L.imm3_query_create.argtypes = [vp, vp, vp, i32, P(CSelect), i32, vp, i32, i64, i32, P(vp)]
    L.imm3_query_create_agg.argtypes = [vp, vp, vp, i32, P(CSelect), i32, vp, i32, vp, i32, i32, P(vp)]
    L.imm3_table_create.argtypes = [vp, P(vp), i32, P(vp)]
    L.imm3_table_destroy.argtypes = [vp]
    L.imm3_query_create_table.argtypes = [vp, vp, vp, i32, P(CSelect), i32, vp, i32, i64, i32, P(vp)]
    L.imm3_query_create_table_agg.argtypes = [vp, vp, vp, i32, P(CSelect), i32, vp, i32, vp, i32, i32, P(vp)]
    L.imm3_query_segment_starts.argtypes = [vp, P(i32), vp, vp]
    L.imm3_query_locate_rows.argtypes = [vp, vp, u64, vp, vp]
    L.imm3_query_group_count.argtypes = [vp, P(C.c_uint32)]
    L.imm3_query_fetch_groups.argtypes = [vp, vp, vp, vp, vp, C.c_uint32]
    L.imm3_query_destroy.argtypes = [vp]
    L.imm3_query_reserve_rows.argtypes = [vp, u64]
    L.imm3_query_run.argtypes = [vp]
    L.imm3_query_run_select.argtypes = [vp]
    L.imm3_query_run_count.argtypes = [vp]
    L.imm3_query_sync.argtypes = [vp]
    L.imm3_query_join_count.argtypes = [vp]
    L.imm3_query_log_counts.argtypes = [vp, vp, C.c_uint64]
    L.imm3_query_layout.argtypes = [vp, P(i32), P(i64), P(i64)]
    L.imm3_query_batches.argtypes = [vp, vp, vp, vp]
    L.imm3_query_count.argtypes = [vp, P(u64)]
    L.imm3_query_bitmap.argtypes = [vp, vp, i64]
    L.imm3_query_row_count.argtypes = [vp, P(u64)]
    L.imm3_query_fetch_rows.argtypes = [vp, vp, P(vp), u64]
    L.imm3_query_device_ptr.argtypes = [vp, i32, P(vp)]
    L.imm3_ctx_timing_enable.argtypes = [vp, i32]
    L.imm3_ctx_timing_reset.argtypes = [vp]
    L.imm3_ctx_timing_mask.argtypes = [vp, C.c_uint32]
    L.imm3_ctx_timing_collect.argtypes = [vp, i32, vp, i32, P(i32)]
    L.imm3_ctx_set_tuning.argtypes = [vp, i32, i32]
    L.imm3_ctx_measure_read_gbps.argtypes = [vp, u64, i32, P(C.c_double)]
    L.imm3_ctx_devclock_enable.argtypes = [vp, i32]
    L.imm3_ctx_devclock_collect.argtypes = [vp, vp, i32, P(i32)]
    L.imm3_query_plan.argtypes = [vp, vp, i32]
    L.imm3_query_agg_shape.argtypes = [vp, P(i32), P(i32), P(i32)]
    L.imm3_ctx_inject_fault.argtypes = [vp, i32, i32, C.c_uint32]
    L.imm3_ctx_debug_device_lock.argtypes = [vp, u64, P(u64)]
    L.imm3_ctx_devclock_raw.argtypes = [vp, i32, vp, i32]
    L.imm3_comm_unique_id.argtypes = [vp]
    L.imm3_comm_create.argtypes = [vp, i32, i32, vp, P(vp)]
    L.imm3_comm_create_all.argtypes = [P(vp), i32, P(vp)]
    L.imm3_comm_destroy.argtypes = [vp]
    L.imm3_comm_info.argtypes = [vp, P(i32), P(i32)]
    L.imm3_comm_sync.argtypes = [vp]
    L.imm3_comm_join.argtypes = [vp]
    L.imm3_comm_allreduce_u64.argtypes = [vp, vp, u64]
    L.imm3_comm_debug_standin.argtypes = [vp, i32, C.c_uint32]
    L.imm3_comm_allreduce_count.argtypes = [vp, P(vp), i32, vp, P(u64)]
    L.imm3_comm_allreduce_count_all.argtypes = [P(vp), i32, P(P(vp)), P(i32), P(u64)]
    L.imm3_comm_merge_groups.argtypes = [vp, P(vp), vp, i32, vp, vp, vp, vp, C.c_uint32, P(C.c_uint32)]
    L.imm3_comm_merge_groups_all.argtypes = [P(vp), i32, P(P(vp)), P(vp), P(i32), vp, vp, vp, vp, C.c_uint32, P(C.c_uint32)]
    for name in EXPORTS + DIAG_EXPORTS:
        fn = getattr(L, name)
        if name not in ("imm3_last_error", "imm3_pfor_encode_bound", "imm3_snappy_encode_bound"):
            fn.restype = C.c_int
    L.imm3_snappy_encode_bound.restype = C.c_uint64
    L.imm3_snappy_encode_bound.argtypes = [C.c_uint64]
    L.imm3_snappy_encode_block.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.imm3_pfor_encode_bound.restype = C.c_uint64
    L.imm3_pfor_encode_bound.argtypes = [C.c_int32]
    L.imm3_pfor_encode_block.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.imm3_pfor_encode_column.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
    _lib = L
    return L


def _check(rc: int):
    if rc != OK:
        msg = load().imm3_last_error()
        raise Imm3Error(rc, msg.decode() if msg else f"imm3 error {rc}")


def pfor_encode_block(values) -> bytes:
    """PFORCodecInt.encode (core/codec/PFORCodec.scala:19-31) of one block of int32 values (host code)."""
    v = np.ascontiguousarray(values, dtype=np.int32)
    cap = load().imm3_pfor_encode_bound(v.size)
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_uint64(0)
    _check(load().imm3_pfor_encode_block(v.ctypes.data, v.size, out.ctypes.data, cap, C.byref(n)))
    return out[:n.value].tobytes()


def pfor_encode_column(values, block_rows: int):
    """A whole int32 column as SegmentWriter would write it with a PFOR_INT codec: (.dat bytes, blockOffset table)."""
    v = np.ascontiguousarray(values, dtype=np.int32)
    nb = (v.size + block_rows - 1) // block_rows
    cap = v.size * 4 + nb * (load().imm3_pfor_encode_bound(0) + 4 * (block_rows // 32 + 4)) + 64
    out = np.empty(cap, dtype=np.uint8)
    offs = np.zeros(nb + 1, dtype=np.int32)
    n = C.c_uint64(0)
    _check(load().imm3_pfor_encode_column(v.ctypes.data, v.size, block_rows, out.ctypes.data, cap, offs.ctypes.data, C.byref(n)))
    return out[:n.value].copy(), offs


def snappy_encode_block(raw) -> bytes:
    """SnappyCodec.encode (core/codec/SnappyCodec.scala:15-27) of one block's raw value bytes (host code)."""
    a = np.frombuffer(bytes(raw), dtype=np.uint8) if not isinstance(raw, np.ndarray) else np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
    cap = load().imm3_snappy_encode_bound(a.size)
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_uint64(0)
    _check(load().imm3_snappy_encode_block(a.ctypes.data, a.size, out.ctypes.data, cap, C.byref(n)))
    return out[:n.value].tobytes()


def device_count() -> int:
    n = C.c_int(0)
    rc = load().imm3_device_count(C.byref(n))
    if rc != OK:
        return 0
    return n.value


class Context:
    """imm3_ctx: device id + HIP stream."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        """stream: a hipStream_t as an int (e.g. torch.cuda.Stream().cuda_stream), or None to let the library create its
        own non-blocking stream.  0 -- what torch reports for its default stream -- cannot be named through the C ABI
        (NULL means "create one"), and silently getting a private stream is how work ends up unordered: refuse it."""
        if stream is not None and int(stream) == 0:
            raise ValueError("stream 0 is HIP's legacy default stream and cannot be passed to imm3_ctx_create; pass None "
                             "(library-owned stream) or a real stream such as torch.cuda.Stream().cuda_stream")
        self._h = C.c_void_p()
        _check(load().imm3_ctx_create(device, C.c_void_p(int(stream)) if stream is not None else None, C.byref(self._h)))
        self.device = device
        # handles created on this context; closed before the context itself (they hold raw pointers into it)
        self._children = weakref.WeakSet()
        self._children_mu = threading.Lock()   # a context may be shared by threads (include/imm3.h, "Threading")

    def _adopt(self, child):
        with self._children_mu:
            self._children.add(child)

    def sync(self):
        _check(load().imm3_ctx_sync(self._h))

    @property
    def stream(self) -> int:
        s = C.c_void_p()
        _check(load().imm3_ctx_stream(self._h, C.byref(s)))
        return s.value or 0

    def capture(self) -> "_Capture":
        """`with ctx.capture() as cap: q0.run(); q1.run()` records the runs (nothing executes); `cap.graph.launch()`
        then enqueues all of their kernels with one call (imm3_ctx_capture_begin / _end, a hipGraph)."""
        return _Capture(self)

    def set_tuning(self, filter_variant: int = 0, grid_blocks: int = 0):
        _check(load().imm3_ctx_set_tuning(self._h, filter_variant, grid_blocks))

    def inject_fault(self, work_group: int = -1, span: int = -1, max_polls: int = 0):
        """Tools' build only (imm3_diag.h): work-group `work_group` of every single-pass launch never announces its `span`-th span;
        look-back waits give up after `max_polls` polls.  (-1, -1, 0) switches it off."""
        _check(load().imm3_ctx_inject_fault(self._h, work_group, span, max_polls))

    def debug_device_lock(self, value: int) -> int:
        """Overwrites the device's single-pass ticket word (0 = free); returns what it held (imm3_diag.h)."""
        prev = C.c_uint64(0)
        _check(load().imm3_ctx_debug_device_lock(self._h, value, C.byref(prev)))
        return prev.value

    def measure_read_gbps(self, nbytes: int = 400_000_000, iters: int = 30) -> float:
        g = C.c_double(0.0)
        _check(load().imm3_ctx_measure_read_gbps(self._h, nbytes, iters, C.byref(g)))
        return g.value

    def devclock_enable(self, max_launches: int):
        _check(load().imm3_ctx_devclock_enable(self._h, max_launches))

    def devclock_collect(self, cap: int = 65536) -> np.ndarray:
        out = np.zeros(cap, dtype=np.float32)
        n = C.c_int32(0)
        _check(load().imm3_ctx_devclock_collect(self._h, out.ctypes.data, cap, C.byref(n)))
        return out[: min(n.value, cap)]

    def devclock_raw(self, launch: int, n: int = 8192) -> np.ndarray:
        out = np.zeros(n, dtype=np.uint64)
        _check(load().imm3_ctx_devclock_raw(self._h, launch, out.ctypes.data, n))
        return out

    def timing_enable(self, max_records: int):
        _check(load().imm3_ctx_timing_enable(self._h, max_records))

    def timing_mask(self, kernel_mask: int):
        _check(load().imm3_ctx_timing_mask(self._h, kernel_mask))

    def timing_reset(self):
        _check(load().imm3_ctx_timing_reset(self._h))

    def timing_collect(self, kernel_id: int, cap: int = 65536) -> np.ndarray:
        out = np.zeros(cap, dtype=np.float32)
        n = C.c_int32(0)
        _check(load().imm3_ctx_timing_collect(self._h, kernel_id, out.ctypes.data, cap, C.byref(n)))
        return out[: min(n.value, cap)]

    def close(self):
        if self._h:
            # The C ABI allows any destruction order (handles are reference counted); closing dependants first simply
            # returns their device memory now instead of when the garbage collector gets to them.
            with self._children_mu:
                kids = list(self._children)
            for kind in (Graph, Comm, DeviceQuery, DeviceTable, DeviceSegment):   # graphs, comms, queries -> tables -> segments
                for k in kids:
                    if isinstance(k, kind):
                        k.close()
            load().imm3_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Graph:
    """imm3_graph: a recorded sequence of query runs."""

    def __init__(self, ctx: Context, handle):
        self.ctx = ctx
        self._h = handle
        ctx._adopt(self)

    def launch(self):
        _check(load().imm3_graph_launch(self._h))

    def close(self):
        if self._h:
            load().imm3_graph_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Capture:
    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.graph: Optional[Graph] = None

    def __enter__(self):
        _check(load().imm3_ctx_capture_begin(self.ctx._h))
        return self

    def __exit__(self, exc_type, exc, tb):
        h = C.c_void_p()
        rc = load().imm3_ctx_capture_end(self.ctx._h, C.byref(h))
        if exc_type is None:
            _check(rc)
            self.graph = Graph(self.ctx, h)
        elif rc == OK:                       # the body failed: drop what was recorded, let its exception through
            load().imm3_graph_destroy(h)
        return False


def _ccolumns(cols):
    """cols: sequence of (codec, width, dat (np.uint8 array or int device ptr), dat_bytes, offsets int32 array)"""
    arr = (CColumn * len(cols))()
    keep = []
    for i, (codec, width, dat, nbytes, offsets) in enumerate(cols):
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        keep.append(offsets)
        arr[i].codec = codec
        arr[i].width = width
        if isinstance(dat, (int, np.integer)):
            arr[i].dat = int(dat)
        else:
            dat = np.ascontiguousarray(dat).view(np.uint8).reshape(-1)
            keep.append(dat)
            arr[i].dat = dat.ctypes.data if dat.size else None
        arr[i].dat_bytes = int(nbytes)
        arr[i].block_offsets = offsets.ctypes.data if offsets.size else None
        arr[i].n_offsets = int(offsets.size)
    return arr, keep


class DeviceSegment:
    """imm3_segment: all columns of one segment id, resident in HBM."""

    def __init__(self, ctx: Context, cols, wrap_device: bool = False, async_copy: bool = False):
        """async_copy: imm3_segment_create_async -- returns once the copies are enqueued on the context's copy stream; the
        host arrays are kept alive here until wait() (or the first close)."""
        self.ctx = ctx
        self.ncols = len(cols)
        arr, keep = _ccolumns(cols)
        self._h = C.c_void_p()
        fn = load().imm3_segment_wrap_device if wrap_device else (load().imm3_segment_create_async if async_copy else load().imm3_segment_create)
        _check(fn(ctx._h, arr, len(cols), C.byref(self._h)))
        ctx._adopt(self)
        self._keep = keep if (wrap_device or async_copy) else None
        self.widths = [c[1] for c in cols]
        self.codecs = [c[0] for c in cols]

    @property
    def device_bytes(self) -> int:
        n = C.c_uint64(0)
        _check(load().imm3_segment_bytes(self._h, C.byref(n)))
        return n.value

    def wait(self):
        """The copies of an async_copy segment have consumed the host buffers."""
        _check(load().imm3_segment_wait(self._h))

    def close(self):
        if self._h:
            load().imm3_segment_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


AGG_COUNT, AGG_MIN, AGG_MAX = 0, 1, 2


class DeviceTable:
    """imm3_table: all segments of one table as one scan unit (one launch over the tile table)."""

    def __init__(self, ctx: Context, segs: Sequence[DeviceSegment]):
        self.ctx, self.segs = ctx, list(segs)
        arr = (C.c_void_p * len(self.segs))(*[s._h for s in self.segs])
        self._h = C.c_void_p()
        _check(load().imm3_table_create(ctx._h, arr, len(self.segs), C.byref(self._h)))
        ctx._adopt(self)
        self.widths, self.codecs = self.segs[0].widths, self.segs[0].codecs

    def close(self):
        if self._h:
            load().imm3_table_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceQuery:
    """imm3_query: ScanOp -> SelectOp* -> ProjectOp (or ProjectAggOp when `aggs` is given) over one device segment."""

    def __init__(self, ctx: Context, seg: DeviceSegment, used_cols: Sequence[int],
                 sels: Sequence[tuple], proj: Sequence[int] = (), limit: int = 0, table_block_size: int = 1024,
                 group_cols: Optional[Sequence[int]] = None, aggs: Optional[Sequence[tuple]] = None):
        self.ctx, self.seg = ctx, seg
        self.used_cols = list(used_cols)
        self.proj = list(proj)
        self.group_cols = list(group_cols or [])
        self.aggs = list(aggs) if aggs is not None else None
        used = np.array(self.used_cols or [0], dtype=np.int32)
        pj = np.array(self.proj or [0], dtype=np.int32)
        cs = (CSelect * max(1, len(sels)))()
        keep = []
        for i, (col, cond, operand) in enumerate(sels):
            cs[i].column = col
            cs[i].cond = cond
            cs[i].value = 0.0
            cs[i].n_match = 0
            if cond in (MATCH, NOTMATCH):
                vals = [bytes(v) for v in (operand or [])]
                blob = np.frombuffer(b"".join(vals) or b"\0", dtype=np.uint8).copy()
                lens = np.array([len(v) for v in vals] or [0], dtype=np.int32)
                keep += [blob, lens]
                cs[i].match_bytes = blob.ctypes.data
                cs[i].match_lens = lens.ctypes.data
                cs[i].n_match = len(vals)
            elif operand is not None:
                cs[i].value = float(operand)
        self._h = C.c_void_p()
        self.is_table = isinstance(seg, DeviceTable)
        create = load().imm3_query_create_table if self.is_table else load().imm3_query_create
        create_agg = load().imm3_query_create_table_agg if self.is_table else load().imm3_query_create_agg
        if self.aggs is not None:
            gc = np.array(self.group_cols or [0], dtype=np.int32)
            ag = np.array([[k, c] for (k, c) in self.aggs] or [[0, 0]], dtype=np.int32)
            _check(create_agg(ctx._h, seg._h, used.ctypes.data, len(self.used_cols), cs, len(sels),
                                                gc.ctypes.data, len(self.group_cols), ag.ctypes.data, len(self.aggs),
                                                table_block_size, C.byref(self._h)))
        else:
            _check(create(ctx._h, seg._h, used.ctypes.data, len(self.used_cols), cs, len(sels),
                          pj.ctypes.data, len(self.proj), limit, table_block_size, C.byref(self._h)))
        ctx._adopt(self)
        nb, tw, nr = C.c_int32(0), C.c_int64(0), C.c_int64(0)
        _check(load().imm3_query_layout(self._h, C.byref(nb), C.byref(tw), C.byref(nr)))
        self.n_batches, self.total_words, self.n_rows = nb.value, tw.value, nr.value
        self.proj_widths = [seg.widths[self.used_cols[j]] for j in self.proj]
        self.proj_codecs = [seg.codecs[self.used_cols[j]] for j in self.proj]

    def batches(self):
        nb = max(self.n_batches, 1)
        size = np.zeros(nb, np.int32)
        oid = np.zeros(nb, np.int32)
        woff = np.zeros(nb, np.int64)
        _check(load().imm3_query_batches(self._h, size.ctypes.data, oid.ctypes.data, woff.ctypes.data))
        return size[: self.n_batches], oid[: self.n_batches], woff[: self.n_batches]

    def segment_starts(self):
        """(first_batch int32[n_seg+1], first_word int64[n_seg+1])"""
        n = C.c_int32(0)
        _check(load().imm3_query_segment_starts(self._h, C.byref(n), None, None))
        fb = np.zeros(n.value + 1, np.int32)
        fw = np.zeros(n.value + 1, np.int64)
        _check(load().imm3_query_segment_starts(self._h, C.byref(n), fb.ctypes.data, fw.ctypes.data))
        return fb, fw

    def locate_rows(self, row_index: np.ndarray):
        """virtual row ids of a table query -> (segment uint32[n], row-in-segment uint32[n])"""
        row_index = np.ascontiguousarray(row_index, dtype=np.uint32)
        seg = np.zeros(max(row_index.size, 1), np.uint32)
        row = np.zeros(max(row_index.size, 1), np.uint32)
        _check(load().imm3_query_locate_rows(self._h, row_index.ctypes.data, row_index.size, seg.ctypes.data, row.ctypes.data))
        return seg[: row_index.size], row[: row_index.size]

    def reserve_rows(self, rows: int):
        _check(load().imm3_query_reserve_rows(self._h, rows))

    def run(self):
        _check(load().imm3_query_run(self._h))

    def run_select(self):
        _check(load().imm3_query_run_select(self._h))

    def run_count(self):
        """ScanOp -> SelectOp* for the count alone: a single-launch chain stores no bitmap."""
        _check(load().imm3_query_run_count(self._h))

    def sync(self):
        _check(load().imm3_query_sync(self._h))

    def join_count(self):
        _check(load().imm3_query_join_count(self._h))

    def log_counts(self, device_ptr: int, capacity: int):
        """Every later run stores its selected-row count at device_ptr[k] (uint64, k = runs since this call)."""
        _check(load().imm3_query_log_counts(self._h, C.c_void_p(device_ptr), C.c_uint64(capacity)))

    def count(self) -> int:
        n = C.c_uint64(0)
        _check(load().imm3_query_count(self._h, C.byref(n)))
        return n.value

    def bitmap(self) -> np.ndarray:
        out = np.zeros(max(self.total_words, 1), dtype=np.uint64)
        _check(load().imm3_query_bitmap(self._h, out.ctypes.data, self.total_words))
        return out[: self.total_words]

    def row_count(self) -> int:
        n = C.c_uint64(0)
        _check(load().imm3_query_row_count(self._h, C.byref(n)))
        return n.value

    def fetch_rows(self):
        """(row_index uint32[n], [per projected column uint8[n, width]])"""
        n = self.row_count()
        cap = max(n, 1)
        idx = np.zeros(cap, dtype=np.uint32)
        cols = [np.zeros((cap, w), dtype=np.uint8) for w in self.proj_widths]
        ptrs = (C.c_void_p * max(1, len(cols)))(*[c.ctypes.data for c in cols])
        _check(load().imm3_query_fetch_rows(self._h, idx.ctypes.data, ptrs, n))
        return idx[:n], [c[:n] for c in cols]

    def fetch_groups(self):
        """(keys uint64[g], first_row uint32[g], counts uint64[g], vals int64[g, n_aggs]) in first-seen order."""
        n = C.c_uint32(0)
        _check(load().imm3_query_group_count(self._h, C.byref(n)))
        g = n.value
        na = max(1, len(self.aggs or []))
        keys = np.zeros(max(g, 1), np.uint64)
        first = np.zeros(max(g, 1), np.uint32)
        counts = np.zeros(max(g, 1), np.uint64)
        vals = np.zeros((max(g, 1), na), np.int64)
        _check(load().imm3_query_fetch_groups(self._h, keys.ctypes.data, first.ctypes.data, counts.ctypes.data, vals.ctypes.data, g))
        return keys[:g], first[:g], counts[:g], vals[:g, : len(self.aggs or [])]

    def plan(self) -> dict:
        """How the library planned this query (include/imm3_diag.h: imm3_query_plan)."""
        v = np.zeros(11, np.int64)
        _check(load().imm3_query_plan(self._h, v.ctypes.data, 11))
        return {"single_pass": bool(v[0]), "P": int(v[1]), "grid": int(v[2]), "spans": int(v[3]), "records": bool(v[4]),
                "rec_dwords": int(v[5]), "ran_single_pass": bool(v[6]), "run_syncs": int(v[7]),
                "abandoned_runs": int(v[8]), "busy_runs": int(v[9]), "limit_gather_gave_up": int(v[10])}

    def device_ptr(self, which: int) -> int:
        p = C.c_void_p()
        _check(load().imm3_query_device_ptr(self._h, which, C.byref(p)))
        return p.value or 0

    def close(self):
        if self._h:
            load().imm3_query_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_predict(n_rows, pred, proj, rec_bytes, sigma, sloc, full):
    """The planner's cost model (csrc/imm3_plan.h): predicted us of plans A, B, C.  pred = [(width, n_match)], proj = [(width, is_pred)]."""
    lib = load()
    pw = (C.c_int32 * max(1, len(pred)))(*[w for w, _ in pred])
    pm = (C.c_int32 * max(1, len(pred)))(*[m for _, m in pred])
    jw = (C.c_int32 * max(1, len(proj)))(*[w for w, _ in proj])
    jp = (C.c_int32 * max(1, len(proj)))(*[1 if p else 0 for _, p in proj])
    out = (C.c_double * 3)()
    lib.imm3_plan_predict.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_void_p]
    lib.imm3_plan_predict.restype = C.c_int
    _check(lib.imm3_plan_predict(n_rows, pw, pm, len(pred), jw, jp, len(proj), rec_bytes, sigma, sloc, full, out))
    return {"A": out[0], "B": out[1], "C": out[2]}


def comm_unique_id() -> bytes:
    """imm3_comm_unique_id: rank 0 makes it, the host hands it to every rank (any channel)."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(load().imm3_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """imm3_comm: this rank's end of the RCCL communicator; the one collective of the path is the count all-reduce."""

    def __init__(self, ctx: Context, world: int, rank: int, unique_id: bytes):
        assert len(unique_id) == COMM_ID_BYTES
        self.ctx, self.world, self.rank = ctx, world, rank
        self._h = C.c_void_p()
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(load().imm3_comm_create(ctx._h, world, rank, buf, C.byref(self._h)))
        ctx._adopt(self)

    @classmethod
    def create_all(cls, ctxs: Sequence[Context]) -> List["Comm"]:
        """Single-process flavour (ncclCommInitAll): one context per device, one Comm per context."""
        n = len(ctxs)
        harr = (C.c_void_p * n)(*[c._h for c in ctxs])
        out = (C.c_void_p * n)()
        _check(load().imm3_comm_create_all(harr, n, out))
        comms = []
        for i, c in enumerate(ctxs):
            o = cls.__new__(cls)
            o.ctx, o.world, o.rank, o._h = c, n, i, C.c_void_p(out[i])
            c._adopt(o)
            comms.append(o)
        return comms

    def sync(self):
        _check(load().imm3_comm_sync(self._h))

    def join(self):
        """The context's stream waits (stream side) for the last collective."""
        _check(load().imm3_comm_join(self._h))

    def debug_standin(self, work_groups: int, spin_us: int):
        """Tools' build: a kernel with RCCL's footprint in front of every count all-reduce (include/imm3_diag.h)."""
        _check(load().imm3_comm_debug_standin(self._h, work_groups, spin_us))

    def allreduce_u64(self, device_ptr: int, n: int):
        _check(load().imm3_comm_allreduce_u64(self._h, C.c_void_p(device_ptr), n))

    def allreduce_count(self, queries: Sequence["DeviceQuery"], device_out: int = 0, wait: bool = True) -> Optional[int]:
        """Sum of the queries' selected-row counts over all ranks.  wait=False only enqueues (device_out receives it)."""
        qs = (C.c_void_p * max(1, len(queries)))(*[q._h for q in queries])
        host = C.c_uint64(0)
        _check(load().imm3_comm_allreduce_count(self._h, qs, len(queries), C.c_void_p(device_out) if device_out else None,
                                                C.byref(host) if wait else None))
        return host.value if wait else None

    def merge_groups(self, queries: Sequence["DeviceQuery"], segment_index: Sequence[int]):
        """ProjectAggregateQueueOp across segments and ranks: (keys uint64[g], first uint64[g] = segment << 32 | row, counts uint64[g],
        vals int64[g, n_aggs]) in first-seen order; every rank gets the whole table."""
        qs = (C.c_void_p * max(1, len(queries)))(*[q._h for q in queries])
        seg = np.ascontiguousarray(segment_index, dtype=np.int32)
        na = max(1, len(queries[0].aggs or [])) if queries else 1
        n = C.c_uint32(0)
        _check(load().imm3_comm_merge_groups(self._h, qs, seg.ctypes.data, len(queries), None, None, None, None, 0, C.byref(n)))
        g = n.value
        keys, first, counts = np.zeros(max(g, 1), np.uint64), np.zeros(max(g, 1), np.uint64), np.zeros(max(g, 1), np.uint64)
        vals = np.zeros((max(g, 1), na), np.int64)
        _check(load().imm3_comm_merge_groups(self._h, qs, seg.ctypes.data, len(queries), keys.ctypes.data, first.ctypes.data, counts.ctypes.data,
                                             vals.ctypes.data, g, C.byref(n)))
        return keys[:g], first[:g], counts[:g], vals[:g]

    @staticmethod
    def merge_groups_all(comms: Sequence["Comm"], queries_per_comm: Sequence[Sequence["DeviceQuery"]], segments_per_comm: Sequence[Sequence[int]]):
        """Single-process flavour of merge_groups: one Comm per device (create_all), each with its queries and their segment indices."""
        n = len(comms)
        carr = (C.c_void_p * n)(*[c._h for c in comms])
        qarrs = [(C.c_void_p * max(1, len(qs)))(*[q._h for q in qs]) for qs in queries_per_comm]
        qq = (C.POINTER(C.c_void_p) * n)(*[C.cast(a, C.POINTER(C.c_void_p)) for a in qarrs])
        segs = [np.ascontiguousarray(s, dtype=np.int32) for s in segments_per_comm]
        ss = (C.c_void_p * n)(*[s.ctypes.data if s.size else None for s in segs])
        nq = (C.c_int32 * n)(*[len(qs) for qs in queries_per_comm])
        na = max(1, len(next(q for qs in queries_per_comm for q in qs).aggs or []))
        g = C.c_uint32(0)
        _check(load().imm3_comm_merge_groups_all(carr, n, qq, ss, nq, None, None, None, None, 0, C.byref(g)))
        m = g.value
        keys, first, counts = np.zeros(max(m, 1), np.uint64), np.zeros(max(m, 1), np.uint64), np.zeros(max(m, 1), np.uint64)
        vals = np.zeros((max(m, 1), na), np.int64)
        _check(load().imm3_comm_merge_groups_all(carr, n, qq, ss, nq, keys.ctypes.data, first.ctypes.data, counts.ctypes.data, vals.ctypes.data, m, C.byref(g)))
        return keys[:m], first[:m], counts[:m], vals[:m]

    @staticmethod
    def allreduce_count_all(comms: Sequence["Comm"], queries_per_comm: Sequence[Sequence["DeviceQuery"]]) -> int:
        n = len(comms)
        carr = (C.c_void_p * n)(*[c._h for c in comms])
        qarrs = [(C.c_void_p * max(1, len(qs)))(*[q._h for q in qs]) for qs in queries_per_comm]
        qq = (C.POINTER(C.c_void_p) * n)(*[C.cast(a, C.POINTER(C.c_void_p)) for a in qarrs])
        nq = (C.c_int32 * n)(*[len(qs) for qs in queries_per_comm])
        host = C.c_uint64(0)
        _check(load().imm3_comm_allreduce_count_all(carr, n, qq, nq, C.byref(host)))
        return host.value

    def close(self):
        if self._h:
            load().imm3_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
