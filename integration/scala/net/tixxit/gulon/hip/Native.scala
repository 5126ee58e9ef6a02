package net.tixxit.gulon.hip

import java.nio.{ByteBuffer, ByteOrder, FloatBuffer}

/**
 * JNI binding of libgulon_hip.so (include/gulon_hip.h) -- the stub a Gulon maintainer adds.
 * NOT compiled in this repository (no JVM toolchain in the build image); tests/test_integration_docs.py keeps this
 * list, integration/jni/gulon_jni.c and the table of INTEGRATION.md identical.
 * Errors arrive as the reference's own exception classes: IllegalArgumentException for a failed `require`,
 * IllegalStateException, UnsupportedOperationException, OutOfMemoryError, RuntimeException (device failure).
 */
object Native {
  System.loadLibrary("gulon_jni") // links libgulon_hip.so

  // Matrix (Matrix.scala:3)
  @native def datasetCreate(data: FloatBuffer, n: Int, d: Int): Long
  @native def datasetDestroy(handle: Long): Unit
  // KMeans (KMeans.scala:188-196, 18-22 / 57-98, 198-226, 100-106, 134-157)
  @native def kmeansInit(ds: Long, from: Int, s: Int, k: Int, seed: Int, centsOut: Array[Float]): Unit
  @native def kmeansAssign(ds: Long, from: Int, s: Int, centroids: Array[Float], k: Int, rngBatch: Int,
                           assignments: Array[Int]): Unit
  @native def kmeansUpdate(ds: Long, from: Int, s: Int, k: Int, assignments: Array[Int], centsOut: Array[Float]): Unit
  @native def kmeansIterate(ds: Long, from: Int, s: Int, centsIn: Array[Float], k: Int, iters: Int,
                            centsOut: Array[Float]): Unit
  @native def kmeansTrain(ds: Long, from: Int, s: Int, k: Int, maxIterations: Int, seed: Int, centsOut: Array[Float],
                          reportInts: Array[Int], reportFloats: Array[Float], maxReports: Int,
                          nReports: Array[Int]): Unit
  // ProductQuantizer (ProductQuantizer.scala:150-153, 25-35)
  @native def pqTrain(ds: Long, m: Int, k: Int, maxIterations: Int, centsOut: Array[Float],
                      reportInts: Array[Int], reportFloats: Array[Float], maxReports: Int,
                      nReports: Array[Int]): Unit
  @native def pqEncode(ds: Long, m: Int, k: Int, cents: Array[Float], codesOut: Array[Byte]): Unit
  // Index (Index.scala:352-383, 385-391, 417-440 + 83-94, 209-229; TopKHeap.scala:44-53)
  @native def prepareQuery(cents: Array[Float], d: Int, m: Int, k: Int, queries: Array[Float], b: Int,
                           tablesOut: Array[Float]): Unit
  @native def indexCreate(codes: Array[Byte], n: Int, d: Int, m: Int, k: Int, cents: Array[Float],
                          rowBase: Int): Long
  @native def indexDestroy(handle: Long): Unit
  @native def indexContextCreate(handle: Long): Long
  @native def indexBatchQuery(handle: Long, queries: Array[Float], b: Int, k: Int, from: Int, until: Int,
                              outIdx: Array[Int], outDist: Array[Float], outCount: Array[Int],
                              outFlags: Array[Int]): Unit
  @native def exactKnn(ds: Long, from: Int, until: Int, queries: Array[Float], b: Int, k: Int, outIdx: Array[Int],
                       outDist: Array[Float], outCount: Array[Int], outFlags: Array[Int]): Unit
  @native def topkMerge(partDist: Array[Float], partIdx: Array[Int], lists: Int, b: Int, k: Int, outIdx: Array[Int],
                        outDist: Array[Float], outCount: Array[Int], outFlags: Array[Int]): Unit
  // PQIndex row-sharded over the GPUs of the node inside this JVM (RCCL all-gathers between the devices)
  @native def shardedIndexCreate(codes: Array[Byte], n: Int, d: Int, m: Int, k: Int, cents: Array[Float],
                                 devices: Array[Int]): Long
  @native def shardedIndexDestroy(handle: Long): Unit
  @native def shardedIndexBatchQuery(handle: Long, queries: Array[Float], b: Int, k: Int, outIdx: Array[Int],
                                     outDist: Array[Float], outCount: Array[Int], outFlags: Array[Int]): Unit
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
