package immutabledb.gpu

import java.nio.ByteBuffer

/**
  * JVM side of the C ABI in include/imm3.h, bound by integration/jni/imm3_jni.c.
  * Handles are opaque pointers carried as Long.  Every method throws java.lang.Exception(msg) when the
  * library reports a non-zero status -- the same convention as the CPU operators
  * (engine/.../operator/Scan.scala:49, Select.scala:22,41,80).
  *
  * Written against the reference at v0; NOT compiled in this repository (no JDK / sbt in the build image): UNVERIFIED.
  */
object Native {
  System.loadLibrary("imm3_jni") // which links libimm3.so

  // CodecType.id order (core/.../codec/Codec.scala:21-24) == IMM3_PFOR_INT .. IMM3_DENSE_STRING
  // SelectCondition codes == IMM3_MATCH .. IMM3_NOOP (include/imm3.h)
  val MATCH = 0; val NOTMATCH = 1; val EQ = 2; val GT = 3; val LT = 4; val NOOP = 5

  @native def ctxCreate(device: Int): Long
  @native def ctxDestroy(ctx: Long): Unit

  /** dats: the direct (mapped) buffers SegmentManager holds; offsets: SegmentMeta.blockOffsets per column. */
  @native def segmentCreate(ctx: Long, codecs: Array[Int], widths: Array[Int],
                            dats: Array[ByteBuffer], offsets: Array[Array[Int]]): Long
  @native def segmentDestroy(seg: Long): Unit

  @native def queryCreate(ctx: Long, seg: Long, usedCols: Array[Int],
                          selCols: Array[Int], selConds: Array[Int], selValues: Array[Double],
                          selMatch: Array[Array[Array[Byte]]],
                          proj: Array[Int], limit: Long, blockSize: Int): Long
  @native def queryDestroy(q: Long): Unit
  @native def queryRun(q: Long): Unit
  @native def queryRunCount(q: Long): Unit   // selected.size summed, no BitSets materialised

  /** packed (size(0..n), oid(0..n), wordOff(0..n)) */
  @native def queryBatches(q: Long): Array[Long]
  /** batch-major selection bitmap; slice [wordOff(k), +ceil(size(k)/64)) is batch k's BitSet words */
  @native def queryBitmap(q: Long): Array[Long]
  @native def queryCount(q: Long): Long
  @native def queryRowCount(q: Long): Long
  @native def queryFetchRows(q: Long, rowIndex: ByteBuffer, cols: Array[ByteBuffer], maxRows: Long): Unit

  // ---- graphs: between captureBegin and captureEnd, queryRun records instead of executing; graphLaunch replays the lot ----
  @native def captureBegin(ctx: Long): Unit
  @native def captureEnd(ctx: Long): Long
  @native def graphLaunch(graph: Long): Unit
  @native def graphDestroy(graph: Long): Unit

  // ---- multi-GPU: one context per device, one communicator per context (imm3_comm_create_all = ncclCommInitAll) ----
  @native def commCreateAll(ctxs: Array[Long]): Array[Long]
  @native def commDestroy(comm: Long): Unit
  /** queries(i): the queries device i ran in this pass.  Selected-row count over all devices: ONE 8-byte
    * ncclAllReduce(sum) per device over RCCL / xGMI -- the only collective of the path (imm3_comm_allreduce_count_all). */
  @native def commAllreduceCountAll(comms: Array[Long], queries: Array[Array[Long]]): Long
  /** ProjectAggregateQueueOp over the devices: [n, keys(n), first(n) = segment << 32 | row, counts(n), vals(n * nAggs)] */
  @native def commMergeGroupsAll(comms: Array[Long], queries: Array[Array[Long]], segments: Array[Array[Int]], nAggs: Int): Array[Long]
}
