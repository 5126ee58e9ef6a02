package net.tixxit.gulon.hip

import java.nio.{ByteBuffer, ByteOrder, FloatBuffer}

/**
 * JNI binding of libgulon_hip.so (include/gulon_hip.h) -- the stub a Gulon maintainer adds.
 * NOT compiled in this repository (no JVM toolchain in the build image).
 */
object Native {
  System.loadLibrary("gulon_jni") // links libgulon_hip.so

  @native def datasetCreate(data: FloatBuffer, n: Int, d: Int): Long
  @native def datasetDestroy(handle: Long): Unit
  @native def pqTrain(ds: Long, m: Int, k: Int, maxIterations: Int, centsOut: Array[Float],
                      reportInts: Array[Int], reportFloats: Array[Float], maxReports: Int,
                      nReports: Array[Int]): Unit
  @native def pqEncode(ds: Long, m: Int, k: Int, cents: Array[Float], codesOut: Array[Byte]): Unit
  @native def indexCreate(codes: Array[Byte], n: Int, d: Int, m: Int, k: Int, cents: Array[Float],
                          rowBase: Int): Long
  @native def indexDestroy(handle: Long): Unit
  @native def indexBatchQuery(handle: Long, queries: Array[Float], b: Int, k: Int, from: Int, until: Int,
                              outIdx: Array[Int], outDist: Array[Float], outCount: Array[Int],
                              outFlags: Array[Int]): Unit
  @native def kmeansAssign(ds: Long, from: Int, s: Int, centroids: Array[Float], k: Int, rngBatch: Int,
                           assignments: Array[Int]): Unit
  // GroupedIndex (Index.scala:231-308); strategy 0 = LimitGroups, 1 = LimitVectors
  @native def groupResiduals(ds: Long, perm: Array[Int], groupOf: Array[Int], centroids: Array[Float],
                             groups: Int): Long
  @native def groupedIndexCreate(codes: Array[Byte], n: Int, d: Int, m: Int, k: Int, pqCents: Array[Float],
                                 groupCents: Array[Float], offsets: Array[Int], groups: Int): Long
  @native def groupedIndexDestroy(handle: Long): Unit
  @native def groupedIndexBatchQuery(handle: Long, queries: Array[Float], b: Int, k: Int, strategy: Int,
                                     limit: Int, outIdx: Array[Int], outDist: Array[Float],
                                     outCount: Array[Int]): Unit

  /** Matrix.data (jagged) -> one direct row-major buffer (Matrix.scala:3). */
  def flatten(rows: Array[Array[Float]], cols: Int): FloatBuffer = {
    val buf = ByteBuffer.allocateDirect(rows.length * cols * 4).order(ByteOrder.nativeOrder()).asFloatBuffer()
    rows.foreach(buf.put(_, 0, cols))
    buf.rewind()
    buf
  }
}
