package immutabledb.operator

import java.nio.{ByteBuffer, ByteOrder}

import immutabledb._
import immutabledb.codec._
import immutabledb.gpu.Native
import immutabledb.storage._

import scala.collection.mutable

/**
  * Drop-in GPU operators for the scan / select / project path.  Same constructor and factory shapes as
  * ScanOp / SelectOp / ProjectOp (Scan.scala:10-17, Select.scala:5-14, Project.scala:8-17), so Engine.execute
  * (engine/.../engine/Engine.scala:167-173) swaps three factory calls:
  *
  *     ScanOp.mkScanOp(sm, query.table)          -> GpuScanOp.mkScanOp(gpus, query.table, cols)   (cols: the SELECT list)
  *     SelectOp.mkSelectOp(col, cond)            -> GpuSelectOp.mkSelectOp(col, cond)              (Engine.scala:125)
  *     ProjectOp.mkProjectOp(cols, limit)        -> GpuProjectOp.mkProjectOp(cols, limit)
  *
  * The operators are plan builders: the first `iterator` call fuses the chain of one segment into one
  * imm3_query -- scan + select + compact + gather on the HBM-resident segment.  PipelineThread (Engine.scala:235-262)
  * is unchanged: it still drains `iterator` and queues FilledColumnVectorBatch objects.
  *
  * What still runs on the JVM, per segment: one array allocation per SELECT-list column per batch and one store per
  * SURVIVING row (filling the vectors ProjectOp reads), the BitSet wrappers, and the queue hand-off.  What no longer
  * runs there: BlockIterator.next's block copy (Segment.scala:162-170), DenseCodec*.decode's per-element
  * read + box + append (DenseCodec.scala:37-73 -- the reference's dominant cost), BitSet.add per row (Scan.scala:55-57)
  * and BitSet.remove per failing row (Select.scala:67-70).  No block of any column is decoded on the CPU.
  *
  * Written against the reference at v0; NOT compiled in this repository (no JDK / sbt in the build image): UNVERIFIED.
  */
/**
  * Threading (include/imm3.h, "Threading"): ONE context per device, shared by every PipelineThread of that device
  * (Engine.scala:176-180: FixedThreadPool(cpuCount), one thread per segment).  The library guards the context's own
  * state; each thread creates, runs, fetches and destroys its OWN imm3_query, which is all a GpuScanOp does.
  */
class GpuSegmentManager(val sm: SegmentManager, val device: Int = 0) {
  val ctx: Long = Native.ctxCreate(device)
  private val segs = mutable.Map[(String, Int), Long]()

  /** All columns of segment `segIdx`, staged into HBM once (SegmentManager keeps the mmaps the same way). */
  def deviceSegment(tableName: String, segIdx: Int): Long = segs.synchronized {
    segs.getOrElseUpdate((tableName, segIdx), {
      val table = sm.getTable(tableName)
      val cols = table.columns
      Native.segmentCreate(ctx,
        cols.map(_.codec.id).toArray,
        cols.map(c => Column.getCodec(c).dtype.size).toArray,
        cols.map(c => sm.segments(s"$tableName.${c.name}")(segIdx)).toArray,
        cols.map(c => sm.segmentsMeta(s"$tableName.${c.name}")(segIdx).blockOffsets).toArray)
    })
  }
}

/**
  * One GpuSegmentManager per device; segment s lives on device s mod G (SURVEY 8e) -- the per-segment pipelines the
  * Engine fans out (Engine.scala:176-180) land on the GPU that holds their segment, nothing is exchanged between GPUs
  * but the selected-row count: countAll() is one 8-byte ncclAllReduce(sum) per device over RCCL / xGMI.
  */
class GpuDevices(val sm: SegmentManager, val nDevices: Int) {
  val managers: Vector[GpuSegmentManager] = (0 until nDevices).map(d => new GpuSegmentManager(sm, d)).toVector
  lazy val comms: Array[Long] = Native.commCreateAll(managers.map(_.ctx).toArray)
  def of(segIdx: Int): GpuSegmentManager = managers(segIdx % nDevices)
  /** queriesPerDevice(d): the imm3_query handles device d ran in this pass */
  def countAll(queriesPerDevice: Array[Array[Long]]): Long =
    if (nDevices == 1) queriesPerDevice(0).map(Native.queryCount).sum
    else Native.commAllreduceCountAll(comms, queriesPerDevice)
  /** ProjectAggregateQueueOp (ProjectAggregateQueue.scala:9-55) across the devices: queriesPerDevice(d) = the aggregation queries
    * device d ran, segmentsPerDevice(d) = the segment index of each; the merged table in first-seen (segment, row) order. */
  def mergeGroups(queriesPerDevice: Array[Array[Long]], segmentsPerDevice: Array[Array[Int]], nAggs: Int): Array[Long] =
    Native.commMergeGroupsAll(comms, queriesPerDevice, segmentsPerDevice, nAggs)
}

object GpuScanOp {
  /** projectCols: the columns the consumer-side ProjectOp will read (the query's SELECT list); Nil = every used column */
  def mkScanOp(gpus: GpuDevices, tableName: String, projectCols: List[String] = Nil) = new Function2[List[Column], Int, GpuScanOp] {
    def apply(cols: List[Column], segIdx: Int) = new GpuScanOp(gpus.of(segIdx), segIdx, tableName, cols, projectCols)
  }
}

class GpuScanOp(val gsm: GpuSegmentManager, val segIdx: Int, val tableName: String, val cols: List[Column],
                val projectCols: List[String] = Nil) extends ColumnVectorOperator {
  val table: Table = gsm.sm.getTable(tableName)

  /** Runs ScanOp -> leaves* -> compact + gather on the GPU and hands out the batches the CPU operators would have produced:
    * `selected` from the GPU bitmap; the column vectors carry values at the SELECTED positions only (the only positions a
    * downstream ProjectOp reads, Project.scala:50-57), filled from the rows the GPU gathered.  No CPU decode. */
  def batches(leaves: List[(String, SelectCondition)]): Iterator[ColumnVectorBatch] = {
    val usedIdx = cols.map(c => table.columns.indexWhere(_.name == c.name)).toArray
    val names = cols.map(_.name)
    val conds = leaves.map(_._2)
    val wanted: List[Int] = (if (projectCols.isEmpty) names else projectCols).map(names.indexOf(_)).filter(_ >= 0)
    val q = Native.queryCreate(gsm.ctx, gsm.deviceSegment(tableName, segIdx), usedIdx,
      leaves.map(l => names.indexOf(l._1)).toArray,
      conds.map {
        case Match(_) => Native.MATCH; case NotMatch(_) => Native.NOTMATCH; case EQ(_) => Native.EQ
        case GT(_) => Native.GT; case LT(_) => Native.LT; case _ => Native.NOOP
      }.toArray,
      conds.map { case EQ(v) => v; case GT(v) => v; case LT(v) => v; case _ => 0.0 }.toArray,
      conds.map { case Match(vs) => vs.map(_.getBytes()).toArray; case NotMatch(vs) => vs.map(_.getBytes()).toArray; case _ => null }.toArray,
      wanted.toArray, 0L, table.blockSize)
    try {
      Native.queryRun(q)
      val packed = Native.queryBatches(q)
      val n = packed.length / 3
      val words = Native.queryBitmap(q)
      // the survivors of the whole segment, in batch order and ascending position: row index + one packed array per wanted column
      val nRows = Native.queryRowCount(q).toInt
      val widths = wanted.map(ci => Column.getCodec(cols(ci)).dtype.size)
      val rowIdx = ByteBuffer.allocateDirect(math.max(1, 4 * nRows)).order(ByteOrder.LITTLE_ENDIAN)
      val bufs = widths.map(w => ByteBuffer.allocateDirect(math.max(1, nRows * w)).order(ByteOrder.LITTLE_ENDIAN)).toArray
      Native.queryFetchRows(q, rowIdx, bufs, nRows.toLong)
      var cursor = 0     // next survivor
      var rowBase = 0L   // segment row of the batch's position 0
      (0 until n).iterator.map { k =>
        val size = packed(k).toInt
        val nw = (size + 63) / 64
        val off = packed(2 * n + k).toInt
        val selected = mutable.BitSet.fromBitMaskNoCopy(java.util.Arrays.copyOfRange(words, off, off + nw))
        val nsel = selected.size
        val vectors: Array[ColumnVector] = cols.zipWithIndex.map { case (c, ci) =>
          val j = wanted.indexOf(ci)
          if (j < 0 || nsel == 0) emptyVector(c)   // never read downstream: not materialised at all
          else c.columnType match {
            case ColumnType.INT =>
              val a = new Array[Int](size)
              var i = 0; while (i < nsel) { a((rowIdx.getInt(4 * (cursor + i)).toLong - rowBase).toInt) = bufs(j).getInt(4 * (cursor + i)); i += 1 }
              IntColumnVector(a)
            case ColumnType.TINYINT =>
              val a = new Array[Byte](size)
              var i = 0; while (i < nsel) { a((rowIdx.getInt(4 * (cursor + i)).toLong - rowBase).toInt) = bufs(j).get(cursor + i); i += 1 }
              TinyIntColumnVector(a)
            case ColumnType.STRING =>
              val w = widths(j)
              val a = new Array[String](size)
              var i = 0
              while (i < nsel) {
                val b = new Array[Byte](w); bufs(j).position((cursor + i) * w); bufs(j).get(b)
                a((rowIdx.getInt(4 * (cursor + i)).toLong - rowBase).toInt) = new String(b)   // DataType.scala:70
                i += 1
              }
              StringColumnVector(a)
          }
        }.toArray
        cursor += nsel
        rowBase += size
        FilledColumnVectorBatch(packed(n + k).toInt, size, vectors, cols.toArray, selected,
          if (leaves.isEmpty) true else selected.nonEmpty)
      }
    } finally Native.queryDestroy(q)
  }

  private def emptyVector(c: Column): ColumnVector = c.columnType match {
    case ColumnType.INT => IntColumnVector(Array.empty[Int])
    case ColumnType.TINYINT => TinyIntColumnVector(Array.empty[Byte])
    case ColumnType.STRING => StringColumnVector(Array.empty[String])
  }

  def iterator = batches(Nil)
}

object GpuSelectOp {
  def mkSelectOp(col: String, cond: SelectCondition) = new Function1[ColumnVectorOperator, GpuSelectOp] {
    def apply(op: ColumnVectorOperator) = new GpuSelectOp(col, cond, op)
  }
}

class GpuSelectOp(val col: String, val cond: SelectCondition, val op: ColumnVectorOperator) extends ColumnVectorOperator {
  /** (scan, leaves in application order) */
  def chain: (GpuScanOp, List[(String, SelectCondition)]) = op match {
    case s: GpuScanOp => (s, List((col, cond)))
    case g: GpuSelectOp => val (s, ls) = g.chain; (s, ls :+ ((col, cond)))
    case _ => throw new Exception("GpuSelectOp must sit on a GpuScanOp / GpuSelectOp chain")
  }

  def iterator = {
    val (scan, leaves) = chain
    leaves.foreach {
      case (_, Match(_) | GT(_) | LT(_) | EQ(_)) =>
      case (_, c) => throw new Exception(s"Unsupported condition: $c") // Select.scala:22
    }
    scan.batches(leaves)
  }
}

object GpuProjectOp {
  def mkProjectOp(cols: List[String], limit: Int = 0) = new Function1[ColumnVectorOperator, ProjectionOperator] {
    // On the consumer thread the upstream is ResultQueueOp (Engine.scala:190-191): batches arrive from all segments, and
    // their vectors already hold the GPU-gathered values at the selected positions (GpuScanOp.batches), so the
    // reference's ProjectOp walks them unchanged -- it only ever reads selected positions of SELECT-list columns.
    def apply(op: ColumnVectorOperator) = new ProjectOp(cols, op, limit)
  }

  /** Fused ScanOp -> SelectOp* -> ProjectOp of ONE segment on the GPU; rows in ascending row order (for a planner that
    * projects per segment instead of on the consumer thread: no batches, no BitSets, no vectors at all). */
  def rowsOf(sel: GpuSelectOp, projCols: List[String], limit: Int): Iterator[Row] = {
    val (scan, leaves) = sel.chain
    val names = scan.cols.map(_.name)
    val usedIdx = scan.cols.map(c => scan.table.columns.indexWhere(_.name == c.name)).toArray
    val conds = leaves.map(_._2)
    val q = Native.queryCreate(scan.gsm.ctx, scan.gsm.deviceSegment(scan.tableName, scan.segIdx), usedIdx,
      leaves.map(l => names.indexOf(l._1)).toArray,
      conds.map { case Match(_) => Native.MATCH; case EQ(_) => Native.EQ; case GT(_) => Native.GT; case LT(_) => Native.LT; case _ => Native.NOOP }.toArray,
      conds.map { case EQ(v) => v; case GT(v) => v; case LT(v) => v; case _ => 0.0 }.toArray,
      conds.map { case Match(vs) => vs.map(_.getBytes()).toArray; case _ => null }.toArray,
      projCols.map(names.indexOf(_)).toArray, limit.toLong, scan.table.blockSize)
    try {
      Native.queryRun(q)
      val n = Native.queryRowCount(q).toInt
      val pcols = projCols.map(c => scan.cols(names.indexOf(c)))
      val widths = pcols.map(c => Column.getCodec(c).dtype.size)
      val bufs = widths.map(w => ByteBuffer.allocateDirect(math.max(1, n * w)).order(ByteOrder.LITTLE_ENDIAN)).toArray
      Native.queryFetchRows(q, null, bufs, n.toLong)
      (0 until n).iterator.map { i =>
        Row.fromSeq(pcols.zipWithIndex.map { case (c, j) =>
          c.columnType match {
            case ColumnType.INT => bufs(j).getInt(4 * i)
            case ColumnType.TINYINT => bufs(j).get(i)
            case ColumnType.STRING =>
              val b = new Array[Byte](widths(j)); bufs(j).position(i * widths(j)); bufs(j).get(b); new String(b)
          }
        })
      }
    } finally Native.queryDestroy(q)
  }
}
