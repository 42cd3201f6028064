"""GPU-backed Operator / ScanOp / SelectOp / ProjectOp / Engine -- the host-side mirror of the reference's
operator interface for the hot path (same names, argument meaning and error behaviour):

    engine/src/main/scala/immutabledb/engine/operator/Operator.scala:14-28   Operator / ColumnVectorOperator / ProjectionOperator
    engine/src/main/scala/immutabledb/engine/operator/Scan.scala:10-73       ScanOp, mkScanOp
    engine/src/main/scala/immutabledb/engine/operator/Select.scala:5-165     SelectOp, mkSelectOp
    engine/src/main/scala/immutabledb/engine/operator/Project.scala:8-81     ProjectOp, mkProjectOp
    engine/src/main/scala/immutabledb/engine/Engine.scala:85-128,158-262     getColumns / resolveSelectOps / execute / PipelineThread
    core/src/main/scala/immutabledb/DataVector.scala:15-48                   ColumnVectorBatch family

The operators are PLAN BUILDERS: composing them costs nothing, and the first `iterator` call fuses the
chain ScanOp -> SelectOp* (-> ProjectOp) of one segment into one imm3_query, i.e. one fused scan+select
kernel (+ offsets scan + compact/gather) on the segment's HBM-resident columns.  Everything that touches
data goes through the C ABI (immutable3_amd/native.py -> libimm3.so); there is no CPU evaluation path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np

from . import native
from .native import Imm3Error
from .query import (EQ, GT, LT, And, Avg, Count, Match, Max, Min, NoOp, NoSelect, NotMatch, Or, Project, ProjectAgg,
                    Query, Select, SelectADT, SelectCondition, Sum)
from .schema import CodecType, Column, Row, Table
from .storage import SegmentManager


# ------------------------------------------------------------------------------------------
# vectors (core/.../DataVector.scala)
# ------------------------------------------------------------------------------------------
class BitSet:
    """scala.collection.mutable.BitSet over the uint64 words the GPU produced (bit i <-> word i>>6, bit i&63)."""

    def __init__(self, words: np.ndarray):
        self.words = words

    def contains(self, i: int) -> bool:
        w = i >> 6
        return w < self.words.size and bool((int(self.words[w]) >> (i & 63)) & 1)

    __call__ = contains
    __contains__ = contains

    @property
    def size(self) -> int:
        return int(np.unpackbits(self.words.view(np.uint8)).sum()) if self.words.size else 0

    @property
    def isEmpty(self) -> bool:
        return not self.words.any()

    def toList(self) -> List[int]:
        if not self.words.size:
            return []
        bits = np.unpackbits(self.words.view(np.uint8), bitorder="little")
        return np.flatnonzero(bits).tolist()

    def __iter__(self):
        return iter(self.toList())


@dataclass
class ColumnVector:                 # IntColumnVector / TinyIntColumnVector / StringColumnVector (:42-48)
    data: np.ndarray


class IntColumnVector(ColumnVector):
    pass


class TinyIntColumnVector(ColumnVector):
    pass


class StringColumnVector(ColumnVector):
    pass


@dataclass
class FilledColumnVectorBatch:      # DataVector.scala:24-31
    oid: int
    size: int
    columnVectors: List[ColumnVector]
    columns: List[Column]
    selected: BitSet
    selectedInUse: bool


ColumnVectorBatch = FilledColumnVectorBatch


def _decode_view(col: Column, raw: np.ndarray) -> ColumnVector:
    """DENSE_* decode is a fixed-width little-endian reinterpretation (DenseCodec.scala:37-73): a view."""
    if col.codec == CodecType.DENSE_INT:
        return IntColumnVector(raw.view("<i4"))
    if col.codec == CodecType.DENSE_TINYINT:
        return TinyIntColumnVector(raw.view(np.int8))
    if col.codec == CodecType.DENSE_STRING:
        return StringColumnVector(raw.reshape(-1, col.width))
    raise Exception(f"No implementation for {col.codec}")


# ------------------------------------------------------------------------------------------
# device-resident SegmentManager
# ------------------------------------------------------------------------------------------
class GpuSegmentManager:
    """What SegmentManager is to the reference (mmap everything once, keep it for the process lifetime,
    SegmentManager.scala:20-23), this is to the GPU path: every segment of every table staged into HBM
    once.  Segment s lives on the context `ctxs[s % len(ctxs)]` (segment-per-GPU sharding)."""

    def __init__(self, sm: SegmentManager, ctxs: Sequence[native.Context] | native.Context | None = None,
                 segment_filter: Optional[Callable[[str, int], bool]] = None):
        if ctxs is None:
            ctxs = [native.Context(0)]
        if isinstance(ctxs, native.Context):
            ctxs = [ctxs]
        self.sm = sm
        self.ctxs = list(ctxs)
        self.tables = sm.tables
        self._segs: Dict[Tuple[str, int], native.DeviceSegment] = {}
        self._tables: Dict[str, Optional[native.DeviceTable]] = {}
        self._filter = segment_filter

    def getTable(self, tableName: str) -> Table:
        return self.sm.getTable(tableName)

    def getTableSegmentCount(self, tableName: str) -> int:
        return self.sm.getTableSegmentCount(tableName)

    def ctx_of(self, segIdx: int) -> native.Context:
        return self.ctxs[segIdx % len(self.ctxs)]

    def owns(self, tableName: str, segIdx: int) -> bool:
        return self._filter is None or self._filter(tableName, segIdx)

    def device_segment(self, tableName: str, segIdx: int) -> native.DeviceSegment:
        key = (tableName, segIdx)
        if key not in self._segs:
            t = self.getTable(tableName)
            cols = []
            for c in t.columns:
                k = f"{tableName}.{c.name}"
                dat = self.sm.segments[k][segIdx]
                meta = self.sm.segmentsMeta[k][segIdx]
                cols.append((CodecType.id_of(c.codec), c.width, dat, dat.size, meta.blockOffsets))
            self._segs[key] = native.DeviceSegment(self.ctx_of(segIdx), cols)
        return self._segs[key]

    def device_table(self, tableName: str) -> Optional[native.DeviceTable]:
        """All owned segments of the table as one scan unit (imm3_table), or None when the table cannot take the
        single-launch path (ragged segments, several contexts): callers then fall back to per-segment pipelines."""
        if tableName in self._tables:
            return self._tables[tableName]
        table = None
        if len(self.ctxs) == 1:
            segs = [s for s in range(self.getTableSegmentCount(tableName)) if self.owns(tableName, s)]
            if segs:
                try:
                    table = native.DeviceTable(self.ctxs[0], [self.device_segment(tableName, s) for s in segs])
                    table.segment_ids = segs
                except Imm3Error as e:
                    if e.code != native.ERR_LAYOUT:
                        raise
        self._tables[tableName] = table
        return table

    def close(self):
        for t in self._tables.values():
            if t is not None:
                t.close()
        self._tables.clear()
        for s in self._segs.values():
            s.close()
        self._segs.clear()


# ------------------------------------------------------------------------------------------
# operators
# ------------------------------------------------------------------------------------------
class Operator:                      # Operator.scala:14-16
    def iterator(self):
        raise NotImplementedError

    def __iter__(self):
        return self.iterator()


class ColumnVectorOperator(Operator):  # Operator.scala:18-20
    pass


class ProjectionOperator(Operator):    # Operator.scala:26-28
    pass


def _cond_spec(cond: SelectCondition):
    if isinstance(cond, Match):
        return native.MATCH, [v.encode("utf-8") if isinstance(v, str) else bytes(v) for v in cond.values]
    if isinstance(cond, NotMatch):
        return native.NOTMATCH, [v.encode("utf-8") if isinstance(v, str) else bytes(v) for v in cond.values]
    if isinstance(cond, GT):
        return native.GT, cond.gt
    if isinstance(cond, LT):
        return native.LT, cond.lt
    if isinstance(cond, EQ):
        return native.EQ, cond.eq
    return native.NOOP, None


class ScanOp(ColumnVectorOperator):
    """Scan.scala:17: ScanOp(sm, segIdx, tableName, cols)."""

    def __init__(self, sm: GpuSegmentManager, segIdx: int, tableName: str, cols: Sequence[Column]):
        self.sm, self.segIdx, self.tableName, self.cols = sm, segIdx, tableName, list(cols)

    @staticmethod
    def mkScanOp(sm: GpuSegmentManager, tableName: str):       # Scan.scala:10-15
        return lambda cols, segIdx: ScanOp(sm, segIdx, tableName, cols)

    # -- plan pieces used by SelectOp / ProjectOp --
    def _table(self) -> Table:
        return self.sm.getTable(self.tableName)

    def _used_indices(self) -> List[int]:
        t = self._table()
        names = [c.name for c in t.columns]
        return [names.index(c.name) for c in self.cols]

    def _query(self, leaves, proj_names: Sequence[str] = (), limit: int = 0) -> native.DeviceQuery:
        t = self._table()
        colnames = [c.name for c in self.cols]
        sels = []
        for (col, cond) in leaves:
            # `vec.columns...filter(_.name == col).head` (Select.scala:60): first used column of that name
            if col not in colnames:
                raise Exception("NoSuchElementException: next on empty iterator")
            code, operand = _cond_spec(cond)
            sels.append((colnames.index(col), code, operand))
        # Project.scala:32-35: vecCols maps each SELECT-list name to its position among the batch columns
        proj = []
        for name in proj_names:
            if name not in colnames:
                raise Exception(f"NoSuchElementException: key not found: {name}")
            proj.append(colnames.index(name))
        seg = self.sm.device_segment(self.tableName, self.segIdx)
        return native.DeviceQuery(seg.ctx, seg, self._used_indices(), sels, proj, limit, t.blockSize)

    def _pfor_decoded(self, c: Column) -> np.ndarray:
        """PFOR_INT column of this segment as the GPU decodes it (one projection without predicates), cached."""
        cache = self.__dict__.setdefault("_pfor_cache", {})
        if c.name not in cache:
            seg = self.sm.device_segment(self.tableName, self.segIdx)
            t = self._table()
            q = native.DeviceQuery(seg.ctx, seg, [[x.name for x in t.columns].index(c.name)], [], [0], 0, t.blockSize)
            q.run()
            _, cols = q.fetch_rows()
            q.close()
            raw = cols[0]
            cache[c.name] = (raw.reshape(-1).view("<i4").copy() if c.codec in CodecType.INT_CODECS else
                             raw.reshape(-1).view(np.int8).copy() if c.codec in CodecType.TINYINT_CODECS else raw.reshape(-1, c.width).copy())
        return cache[c.name]

    def _host_vectors(self, k: int, start_row: int, size: int) -> List[ColumnVector]:
        out = []
        for c in self.cols:
            if c.codec == CodecType.PFOR_INT or c.codec in CodecType.SNAPPY:   # compressed: the GPU's decode
                dec = self._pfor_decoded(c)
                if c.codec in CodecType.INT_CODECS:
                    out.append(IntColumnVector(dec[start_row:start_row + size]))
                elif c.codec in CodecType.TINYINT_CODECS:
                    out.append(TinyIntColumnVector(dec[start_row:start_row + size]))
                else:
                    out.append(StringColumnVector(dec[start_row:start_row + size]))
                continue
            dat = self.sm.sm.segments[f"{self.tableName}.{c.name}"][self.segIdx]
            out.append(_decode_view(c, np.asarray(dat[start_row * c.width: (start_row + size) * c.width])))
        return out

    def _batches(self, leaves) -> Iterator[FilledColumnVectorBatch]:
        q = self._query(leaves)
        q.run_select()
        size, oid, woff = q.batches()
        words = q.bitmap()
        q.close()
        row = 0
        for k in range(q.n_batches):
            n = int(size[k])
            nw = (n + 63) // 64
            sel = BitSet(words[int(woff[k]): int(woff[k]) + nw])
            # selectedInUse = false when a SelectOp leaves the batch empty (Select.scala:44-47); ScanOp alone yields true
            in_use = (not sel.isEmpty) if leaves else True
            yield FilledColumnVectorBatch(int(oid[k]), n, self._host_vectors(k, row, n), list(self.cols), sel, in_use)
            row += n

    def iterator(self):
        return self._batches([])


class SelectOp(ColumnVectorOperator):
    """Select.scala:14: SelectOp(col, cond, op).  NotMatch / NoOp are rejected when the iterator is built (:22)."""

    def __init__(self, col: str, cond: SelectCondition, op: ColumnVectorOperator):
        self.col, self.cond, self.op = col, cond, op

    @staticmethod
    def mkSelectOp(col: str, cond: SelectCondition):            # Select.scala:5-12
        return lambda op: SelectOp(col, cond, op)

    def _chain(self):
        """-> (ScanOp, [(col, cond)] in application order: innermost SelectOp first)."""
        leaves = []
        op = self
        while isinstance(op, SelectOp):
            leaves.append((op.col, op.cond))
            op = op.op
        if not isinstance(op, ScanOp):
            raise Exception("SelectOp chain must end in a ScanOp for the fused GPU path")
        leaves.reverse()
        return op, leaves

    def iterator(self):
        scan, leaves = self._chain()
        for (_, cond) in leaves:
            if not isinstance(cond, (Match, GT, LT, EQ)):
                raise Exception(f"Unsupported condition: {cond}")   # Select.scala:22
        return scan._batches(leaves)


class ProjectOp(ProjectionOperator):
    """Project.scala:17: ProjectOp(cols, op, limit = 0)."""

    def __init__(self, cols: Sequence[str], op: ColumnVectorOperator, limit: int = 0):
        self.cols, self.op, self.limit = list(cols), op, limit

    @staticmethod
    def mkProjectOp(cols: Sequence[str], limit: int = 0):       # Project.scala:8-15
        return lambda op: ProjectOp(cols, op, limit)

    def _fused(self):
        if isinstance(self.op, ScanOp):
            return self.op, []
        if isinstance(self.op, SelectOp):
            return self.op._chain()
        return None

    def run_columns(self):
        """Fused execution; returns (row_index uint32[n], [typed numpy array per SELECT-list column])."""
        scan, leaves = self._fused()
        for (_, cond) in leaves:
            if not isinstance(cond, (Match, GT, LT, EQ)):
                raise Exception(f"Unsupported condition: {cond}")
        q = scan._query(leaves, self.cols, self.limit)
        q.run()
        idx, cols = q.fetch_rows()
        out = []
        for raw, codec in zip(cols, q.proj_codecs):
            if codec in native.INT_CODECS:
                out.append(raw.reshape(-1).view("<i4"))
            elif codec in native.TINYINT_CODECS:
                out.append(raw.reshape(-1).view(np.int8))
            else:
                out.append(raw)
        q.close()
        return idx, out

    def iterator(self) -> Iterator[Row]:
        if self._fused() is None:
            return self._host_iterator()
        _, cols = self.run_columns()
        return _rows_from_columns(cols)

    def _host_iterator(self) -> Iterator[Row]:
        """ProjectIterator over batches from a non-fusable upstream (e.g. a queue of batches from several
        segments, Engine.scala:190-191).  Row materialisation is Project.scala:50-63; batches without
        survivors are skipped (the reference faults on them)."""
        total = 0
        for vec in self.op.iterator():
            names = [c.name for c in vec.columns]
            vec_cols = [names.index(c) for c in self.cols]
            for pos in vec.selected.toList():
                if self.limit > 0 and total >= self.limit:
                    return
                yield Row(*[_value(vec.columnVectors[j].data[pos]) for j in vec_cols])
                total += 1
            if self.limit > 0 and total >= self.limit:
                return


# ------------------------------------------------------------------------------------------
# aggregation (engine/.../operator/ProjectAggregate.scala, ProjectAggregateQueue.scala)
# ------------------------------------------------------------------------------------------
def java_double_to_string(v: float) -> str:
    """Double.toString for the integral values Max/MinDoubleAggr hold (value.toDouble of an Int / Byte)."""
    iv = int(v)
    if abs(iv) < 10 ** 7:
        return f"{iv}.0"
    digits = str(abs(iv))
    return ("-" if iv < 0 else "") + digits[0] + "." + (digits[1:].rstrip("0") or "0") + "E" + str(len(digits) - 1)


class Aggregator:                    # ProjectAggregate.scala:11-20
    kind = native.AGG_COUNT

    def __init__(self, col: str, alias: str):
        self.col, self.alias = col, alias

    def make(self):
        return type(self)(self.col, self.alias)


class CountAggr(Aggregator):         # :22-35
    kind = native.AGG_COUNT

    def __init__(self, col, alias):
        super().__init__(col, alias)
        self.counter = 0

    def set(self, n):
        self.counter = n

    def get(self):
        return self.counter

    def combine(self, other):
        self.counter += other.get()
        return self

    def repr(self):
        return str(self.counter)


class MaxDoubleAggr(Aggregator):     # :37-48
    kind = native.AGG_MAX

    def __init__(self, col, alias):
        super().__init__(col, alias)
        self.value = -1.7976931348623157e308

    def add(self, v):
        if v > self.value:
            self.value = float(v)

    def get(self):
        return self.value

    def combine(self, other):
        self.add(other.get())
        return self

    def repr(self):
        return java_double_to_string(self.value)


class MinDoubleAggr(MaxDoubleAggr):  # :50-61
    kind = native.AGG_MIN

    def __init__(self, col, alias):
        Aggregator.__init__(self, col, alias)
        self.value = 1.7976931348623157e308

    def add(self, v):
        if v < self.value:
            self.value = float(v)


class MaxStringAggr(Aggregator):     # :79-91
    kind = native.AGG_MAX

    def __init__(self, col, alias):
        super().__init__(col, alias)
        self.value = ""

    def add(self, v):
        if self.value == "" or v > self.value:
            self.value = v

    def get(self):
        return self.value

    def combine(self, other):
        self.add(other.get())
        return self

    def repr(self):
        return self.value


def resolveProjectOp(projectAgg: ProjectAgg, table: Table):
    """Engine.resolveProjectOp (Engine.scala:130-156): Aggregate ADT -> Aggregators, default aliases
    col_max / col_min / col_count; Min over a STRING column becomes MaxStringAggr (the reference's own mapping,
    :145); Sum / Avg parse but are rejected (:152)."""
    aggs = []
    for a in projectAgg.aggs:
        if isinstance(a, Max):
            ct = table.getColumn(a.col).columnType
            aggs.append((MaxStringAggr if ct == "STRING" else MaxDoubleAggr)(a.col, a.alias or a.col + "_max"))
        elif isinstance(a, Min):
            ct = table.getColumn(a.col).columnType
            aggs.append((MaxStringAggr if ct == "STRING" else MinDoubleAggr)(a.col, a.alias or a.col + "_min"))
        elif isinstance(a, Count):
            aggs.append(CountAggr(a.col, a.alias or a.col + "_count"))
        else:
            raise Exception("Unknown Aggregate type")
    return lambda op: ProjectAggOp(aggs, op, list(projectAgg.groupBy))


class ProjectAggOp(Operator):
    """ProjectAggOp(aggs, op, groupBy) (ProjectAggregate.scala:125): yields (groupKey, {alias: Aggregator}) in
    first-seen order.  The whole chain of one segment runs fused on the GPU: scan+select kernel, then the LDS
    hash-aggregation kernel; group keys and aggregates come back already reduced."""

    def __init__(self, aggs: Sequence[Aggregator], op: ColumnVectorOperator, groupBy: Sequence[str]):
        self.aggs, self.op, self.groupBy = list(aggs), op, list(groupBy)

    @staticmethod
    def make(aggs, groupBy):           # ProjectAggregate.scala:116-122
        return lambda op: ProjectAggOp(aggs, op, groupBy)

    def iterator(self):
        if isinstance(self.op, ScanOp):
            scan, leaves = self.op, []
        else:
            scan, leaves = self.op._chain()
        for (_, cond) in leaves:
            if not isinstance(cond, (Match, GT, LT, EQ)):
                raise Exception(f"Unsupported condition: {cond}")
        colnames = [c.name for c in scan.cols]
        # groupCols filters the BATCH columns by membership in groupBy (:135-140): batch-column order
        group_idx = [i for i, n in enumerate(colnames) if n in self.groupBy]
        by_alias = {}
        for a in self.aggs:              # aggsMap is a HashMap keyed by alias: a later duplicate replaces an earlier one
            by_alias[a.alias] = a
        aggs = list(by_alias.values())
        t = scan._table()
        sels = []
        for (col, cond) in leaves:
            code, operand = _cond_spec(cond)
            sels.append((colnames.index(col), code, operand))
        seg = scan.sm.device_segment(scan.tableName, scan.segIdx)
        yield from self._run(seg, scan.cols, scan._used_indices(), sels, t.blockSize)

    def _run(self, seg, cols, used_idx, sels, block_size):
        """seg: a DeviceSegment or a DeviceTable (then the groups are already merged across segments)."""
        colnames = [c.name for c in cols]
        group_idx = [i for i, n in enumerate(colnames) if n in self.groupBy]
        by_alias = {}
        for a in self.aggs:
            by_alias[a.alias] = a
        aggs = list(by_alias.values())

        class _S:                             # the per-row decoding below only needs `cols`
            pass
        scan = _S()
        scan.cols = list(cols)
        q = native.DeviceQuery(seg.ctx, seg, used_idx, sels, (), 0, block_size,
                               group_cols=group_idx, aggs=[(a.kind, colnames.index(a.col)) for a in aggs])
        q.run()
        keys, first, counts, vals = q.fetch_groups()
        q.close()
        widths = [scan.cols[i].width for i in group_idx]
        for g in range(keys.shape[0]):
            raw = int(keys[g]).to_bytes(8, "little")
            parts, off = [], 0
            for i, w in zip(group_idx, widths):
                parts.append(_key_part(scan.cols[i], raw[off: off + w]))
                off += w
            out = {}
            for j, a in enumerate(aggs):
                na = a.make()
                if isinstance(na, CountAggr):
                    na.set(int(counts[g]))
                elif isinstance(na, MaxStringAggr):
                    w = scan.cols[colnames.index(a.col)].width
                    na.value = int(vals[g, j]).to_bytes(8, "big", signed=True)[8 - w:].decode("utf-8", errors="replace")
                else:
                    na.value = float(int(vals[g, j]))
                out[a.alias] = na
            yield "_".join(parts), out


def _key_part(col: Column, raw: bytes) -> str:
    if col.codec in CodecType.INT_CODECS:
        return str(int.from_bytes(raw, "little", signed=True))
    if col.codec in CodecType.TINYINT_CODECS:
        return str(int.from_bytes(raw, "little", signed=True))
    return raw.decode("utf-8", errors="replace")


def _value(x):
    if isinstance(x, np.ndarray):           # fixed-width string: new String(bytes) (DataType.scala:70)
        return bytes(x).decode("utf-8", errors="replace")
    return int(x)


def _rows_from_columns(cols: List[np.ndarray]) -> Iterator[Row]:
    n = cols[0].shape[0] if cols else 0
    for i in range(n):
        yield Row(*[_value(c[i]) for c in cols])


# ------------------------------------------------------------------------------------------
# Engine (the caller of the hot path; only what the path needs: planning + per-segment fan-out)
# ------------------------------------------------------------------------------------------
def _column_hash(c: Column) -> int:
    from . import scala_sets
    codec_id = CodecType.id_of(c.codec)
    return scala_sets.product_hash([scala_sets.java_string_hash(c.name), scala_sets.COLUMN_TYPE_ID[c.columnType], codec_id,
                                    scala_sets.map_hash(dict(c.dtypeAttrs))])


def getColumns(query: Query, table: Table) -> List[Column]:
    """Engine.getColumns (Engine.scala:85-106): (rec(query.select).toList ++ projectColumns).toSet.toList.
    Scala's Set1..Set4 keep insertion order, so for <= 4 distinct columns the order is first-seen order
    (SURVEY A.1 rule 3).  From the fifth distinct column on the set is an immutable.HashSet and the order is its hash
    trie's: restated in scala_sets.py (scala-library 2.12.11; unpinned at the reference boundary, pinned against the
    library's well-known Set(1 to 10) order).  The first column of the result defines the batches (Scan.scala:55); for
    aggregations the order also decides how the group key is joined (ProjectAggregate.scala:135-156)."""
    from .scala_sets import ScalaSet

    def rec(sel: SelectADT) -> ScalaSet:
        out = ScalaSet()
        if isinstance(sel, (And, Or)):                 # rec(a) ++ rec(b): b's elements, in b's order, added to a
            out = rec(sel.op1)
            for c in rec(sel.op2).to_list():
                out.add(c, _column_hash(c))
        elif isinstance(sel, Select):
            c = table.getColumn(sel.col)
            out.add(c, _column_hash(c))
        return out

    cols = ScalaSet()
    for c in rec(query.select).to_list():
        cols.add(c, _column_hash(c))
    if isinstance(query.project, Project):
        names = list(query.project.cols)
    else:                                              # ProjectAgg: aggsCols ++ groupCols (Engine.scala:96-100)
        names = [a.col for a in query.project.aggs] + list(query.project.groupBy)
    for name in names:
        c = table.getColumn(name)
        cols.add(c, _column_hash(c))
    return cols.to_list()


def resolveSelectOps(query: Query) -> List[Callable[[ColumnVectorOperator], ColumnVectorOperator]]:
    """Engine.resolveSelectOps + PipelineThread.runOps (Engine.scala:108-128, 237-245) flattened: the PTree is
    folded left-to-right, PNode(n1, n2, _) => rec(n2) o rec(n1), and the AND/OR tag is ignored."""
    def rec(sel: SelectADT):
        if isinstance(sel, (And, Or)):
            return rec(sel.op1) + rec(sel.op2)
        if isinstance(sel, Select):
            return [SelectOp.mkSelectOp(sel.col, sel.cond)]
        return []                                             # NoSelect => identity
    return rec(query.select)


class Engine:
    """Engine.execute (Engine.scala:158-197) for Project queries.  One fused pipeline per segment (the
    reference: one PipelineThread per segment, :176-180); output order across segments is unspecified in
    the reference (queue interleaving, :255) and defined here as ascending segment index."""

    def __init__(self, sm: GpuSegmentManager):
        self.sm = sm

    def pipelines(self, query: Query):
        table = self.sm.getTable(query.table)
        used = getColumns(query, table)
        leaves = resolveSelectOps(query)
        mk_scan = ScanOp.mkScanOp(self.sm, query.table)
        mk_proj = ProjectOp.mkProjectOp(list(query.project.cols), query.project.limit)
        for segIdx in range(self.sm.getTableSegmentCount(table.name)):
            if not self.sm.owns(table.name, segIdx):
                continue
            op: ColumnVectorOperator = mk_scan(used, segIdx)
            for leaf in leaves:
                op = leaf(op)
            yield segIdx, mk_proj(op)

    def _table_plan(self, query: Query):
        """(DeviceTable, used columns, select specs) when the whole table can run as ONE fused launch, else None."""
        table = self.sm.getTable(query.table)
        dt = self.sm.device_table(query.table)
        if dt is None:
            return None
        used = getColumns(query, table)
        names = [c.name for c in used]
        sels = []
        for leaf in resolveSelectOps(query):
            op = leaf(None)
            if not isinstance(op.cond, (Match, GT, LT, EQ)):
                raise Exception(f"Unsupported condition: {op.cond}")
            code, operand = _cond_spec(op.cond)
            col = used[names.index(op.col)]
            if code == native.MATCH and (col.codec not in CodecType.STRING_CODECS or col.width != 2 or not (0 < len(operand) <= 8)):
                return None                      # the tile kernels take 2-byte strings with <= 8 IN-list values
            sels.append((names.index(op.col), code, operand))
        tnames = [c.name for c in table.columns]
        return dt, used, [tnames.index(n) for n in names], sels

    def execute_agg(self, query: Query):
        """ProjectAgg queries: per-segment ProjectAggOp, then ProjectAggregateQueueOp's combine by group key
        (first arrival first; segments in ascending order).  Returns an ordered dict key -> {alias: Aggregator}.
        When the table qualifies, the whole thing is one imm3 table query (groups already merged on the GPU)."""
        table = self.sm.getTable(query.table)
        plan = self._table_plan(query)
        if plan is not None:
            dt, used, used_idx, sels = plan
            agg_op = resolveProjectOp(query.project, table)(ScanOp(self.sm, 0, query.table, used))
            return dict(agg_op._run(dt, used, used_idx, sels, table.blockSize))
        used = getColumns(query, table)
        leaves = resolveSelectOps(query)
        mk_scan = ScanOp.mkScanOp(self.sm, query.table)
        mk_agg = resolveProjectOp(query.project, table)
        result = {}
        for segIdx in range(self.sm.getTableSegmentCount(table.name)):
            if not self.sm.owns(table.name, segIdx):
                continue
            op = mk_scan(used, segIdx)
            for leaf in leaves:
                op = leaf(op)
            for key, aggmap in mk_agg(op).iterator():
                cur = result.get(key)
                if cur is None:
                    result[key] = aggmap
                else:
                    for alias, agg in aggmap.items():
                        cur[alias] = cur[alias].combine(agg) if alias in cur else agg
        return result

    def execute_table_columns(self, query: Query):
        """Project query over the whole table as ONE fused launch: (segment uint32[n], row uint32[n], [column arrays])
        in ascending (segment, row) order with the global limit applied -- or None when the table does not qualify."""
        plan = self._table_plan(query)
        if plan is None:
            return None
        dt, used, used_idx, sels = plan
        names = [c.name for c in used]
        proj = [names.index(n) for n in query.project.cols]
        q = native.DeviceQuery(dt.ctx, dt, used_idx, sels, proj, query.project.limit, self.sm.getTable(query.table).blockSize)
        q.run()
        idx, cols = q.fetch_rows()
        seg, row = q.locate_rows(idx)
        out = []
        for raw, codec in zip(cols, q.proj_codecs):
            out.append(raw.reshape(-1).view("<i4") if codec in native.INT_CODECS else
                       raw.reshape(-1).view(np.int8) if codec in native.TINYINT_CODECS else raw)
        q.close()
        seg_ids = np.asarray(getattr(dt, "segment_ids", list(range(len(dt.segs)))), dtype=np.uint32)
        return seg_ids[seg] if seg.size else seg, row, out

    def execute(self, query: Query) -> Iterator[Row]:
        if isinstance(query.project, ProjectAgg):
            # ProjectAggregateQueueOp.next: Row of the aggregators' repr.  The reference lists them in the iteration
            # order of a mutable.HashMap keyed by alias (not reproduced); here: SELECT-list order.
            for _, aggmap in self.execute_agg(query).items():
                yield Row(*[a.repr() for a in aggmap.values()])
            return
        fused = self.execute_table_columns(query)
        if fused is not None:
            yield from _rows_from_columns(fused[2])
            return
        limit = query.project.limit
        total = 0
        for _, proj in self.pipelines(query):
            for row in proj.iterator():
                if limit > 0 and total >= limit:
                    return
                yield row
                total += 1

    def execute_columns(self, query: Query):
        """Columnar result per segment: [(segIdx, row_index, [column arrays])] (no Row boxing)."""
        return [(segIdx, *proj.run_columns()) for segIdx, proj in self.pipelines(query)]
