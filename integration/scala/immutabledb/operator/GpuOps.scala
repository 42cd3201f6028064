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
  *     ScanOp.mkScanOp(sm, query.table)          -> GpuScanOp.mkScanOp(gsm, query.table)
  *     SelectOp.mkSelectOp(col, cond)            -> GpuSelectOp.mkSelectOp(col, cond)      (Engine.scala:125)
  *     ProjectOp.mkProjectOp(cols, limit)        -> GpuProjectOp.mkProjectOp(cols, limit)
  *
  * The operators are plan builders: the first `iterator` call fuses the chain of one segment into one
  * imm3_query (one fused scan+select kernel on the HBM-resident segment).  PipelineThread (Engine.scala:235-262)
  * is unchanged: it still drains `iterator` and queues FilledColumnVectorBatch objects.
  *
  * Written against the reference at v0; NOT compiled in this repository (no JDK / sbt in the build image).
  */
class GpuSegmentManager(val sm: SegmentManager, device: Int = 0) {
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

object GpuScanOp {
  def mkScanOp(gsm: GpuSegmentManager, tableName: String) = new Function2[List[Column], Int, GpuScanOp] {
    def apply(cols: List[Column], segIdx: Int) = new GpuScanOp(gsm, segIdx, tableName, cols)
  }
}

class GpuScanOp(val gsm: GpuSegmentManager, val segIdx: Int, val tableName: String, val cols: List[Column])
    extends ColumnVectorOperator {
  val table: Table = gsm.sm.getTable(tableName)

  /** Runs ScanOp -> leaves* on the GPU and re-materialises the batches the CPU operators would have produced. */
  def batches(leaves: List[(String, SelectCondition)]): Iterator[ColumnVectorBatch] = {
    val usedIdx = cols.map(c => table.columns.indexWhere(_.name == c.name)).toArray
    val names = cols.map(_.name)
    val conds = leaves.map(_._2)
    val q = Native.queryCreate(gsm.ctx, gsm.deviceSegment(tableName, segIdx), usedIdx,
      leaves.map(l => names.indexOf(l._1)).toArray,
      conds.map {
        case Match(_) => Native.MATCH; case NotMatch(_) => Native.NOTMATCH; case EQ(_) => Native.EQ
        case GT(_) => Native.GT; case LT(_) => Native.LT; case _ => Native.NOOP
      }.toArray,
      conds.map { case EQ(v) => v; case GT(v) => v; case LT(v) => v; case _ => 0.0 }.toArray,
      conds.map { case Match(vs) => vs.map(_.getBytes()).toArray; case NotMatch(vs) => vs.map(_.getBytes()).toArray; case _ => null }.toArray,
      Array[Int](), 0L, table.blockSize)
    try {
      Native.queryRun(q)
      val packed = Native.queryBatches(q)
      val n = packed.length / 3
      val words = Native.queryBitmap(q)
      val iters = cols.map(c => gsm.sm.getSegments(tableName, c.name)(segIdx).iterator).toVector
      (0 until n).iterator.map { k =>
        val size = packed(k).toInt
        val nw = (size + 63) / 64
        val off = packed(2 * n + k).toInt
        val selected = mutable.BitSet.fromBitMaskNoCopy(java.util.Arrays.copyOfRange(words, off, off + nw))
        // column vectors for consumers that read values (ProjectOp on the consumer thread): CPU decode of the block
        val vectors: Array[ColumnVector] = cols.zipWithIndex.map { case (c, ci) =>
          val bytes = new java.io.ByteArrayInputStream(iters(ci).next)
          Column.getCodec(c) match {
            case DenseCodecInt => IntColumnVector(DenseCodecInt.decode(bytes))
            case DenseCodecTinyInt => TinyIntColumnVector(DenseCodecTinyInt.decode(bytes))
            case s: DenseCodecString => StringColumnVector(s.decode(bytes))
            case other => throw new Exception(s"No implementation for $other")
          }
        }.toArray
        FilledColumnVectorBatch(packed(n + k).toInt, size, vectors, cols.toArray, selected,
          if (leaves.isEmpty) true else selected.nonEmpty)
      }
    } finally Native.queryDestroy(q)
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
    // On the consumer thread the upstream is ResultQueueOp (Engine.scala:190-191): batches arrive from all
    // segments, so row materialisation stays ProjectOp's.  Per-segment fused projection (compact + gather on the
    // GPU) is GpuProjectOp.rowsOf below, for a planner that projects per segment.
    def apply(op: ColumnVectorOperator) = new ProjectOp(cols, op, limit)
  }

  /** Fused ScanOp -> SelectOp* -> ProjectOp of ONE segment on the GPU; rows in ascending row order. */
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
