package net.tixxit.gulon.hip

import java.util.{Collections, WeakHashMap}

import cats.effect.{ContextShift, IO}

import net.tixxit.gulon.{EncodedMatrix, Index, KeyIndex, MathUtils, Matrix, Metric, ProductQuantizer, TopKHeap, WordVectors}

/**
 * Replacement bodies of the flat index's query path (Index.scala).  The reference keeps its types and signatures and
 * delegates:
 * {{{
 *   // object Index
 *   def sorted(wordVectors: WordVectors.Sorted, quantizer: ProductQuantizer, metric: Metric)
 *             (implicit cs: ContextShift[IO]): IO[Index.SortedIndex] = hip.HipIndex.sorted(wordVectors, quantizer, metric) // :107-114
 *   def prepareQuery(pq: ProductQuantizer, queries: Array[Array[Float]]): PreparedQuery = hip.HipIndex.prepareQuery(pq, queries) // :352-383
 *   def exactNearestNeighbours(vectors: Array[Array[Float]], from: Int, until: Int, query: Array[Float], k: Int): TopKHeap =
 *     hip.HipIndex.exactNearestNeighbours(vectors, from, until, query, k)                                                   // :209-229
 *   // case class PQIndex
 *   def batchQuery(k: Int, vectors: Matrix, from: Int, until: Int): Vector[TopKHeap] = hip.HipIndex.batchQuery(this, k, vectors, from, until) // :417-440
 *   // case class SortedIndex
 *   def batchQuery(k: Int, vectors: Matrix): Vector[Index.Result] = hip.HipIndex.sortedBatchQuery(this, k, vectors)          // :334-337
 * }}}
 * `PQIndex.decode`, `SortedIndex.lookup/query`, `Result`, `Result.fromHeap` stay as they are.
 * NOT compiled in this repository (no JVM toolchain in the build image).
 */
object HipIndex {
  /** One device index per PQIndex VALUE (a case class: equal indexes share the handle), freed with it. */
  final class DeviceIndex(val handle: Long) {
    override def finalize(): Unit = Native.indexDestroy(handle)
  }
  private val handles = Collections.synchronizedMap(new WeakHashMap[EncodedMatrix, DeviceIndex]())

  /** PQIndex(productQuantizer, data) on the device (Index.scala:385-391): codes + codebooks uploaded once. */
  def device(index: Index.PQIndex): DeviceIndex = handles.synchronized {
    val hit = handles.get(index.data)
    if (hit != null) hit
    else {
      val pq = index.productQuantizer
      val packed = index.data.unwrappedEncodings.iterator.flatten.toArray      // m arrays of bytesPerCode, back to back
      val dev = new DeviceIndex(Native.indexCreate(if (packed.isEmpty) new Array[Byte](1) else packed, index.length,
        pq.dimension, pq.quantizers.size, pq.numClusters, HipProductQuantizer.flatCentroids(pq), 0))
      handles.put(index.data, dev)
      dev
    }
  }

  private def flatten(rows: Array[Array[Float]], cols: Int): Array[Float] = {
    val out = new Array[Float](math.max(rows.length * cols, 1))
    var i = 0
    while (i < rows.length) { System.arraycopy(rows(i), 0, out, i * cols, cols); i += 1 }
    out
  }

  /** The raw result of one batch: per query `count` (<= k) rows ascending by distance -- what Result.fromHeap emits. */
  final case class Batch(k: Int, idx: Array[Int], dist: Array[Float], count: Array[Int], flags: Array[Int]) {
    def rows(q: Int): Array[Int] = java.util.Arrays.copyOfRange(idx, q * k, q * k + count(q))
    def distances(q: Int): Array[Float] = java.util.Arrays.copyOfRange(dist, q * k, q * k + count(q))
  }

  /**
   * PQIndex#batchQuery(k, vectors, from, until) (Index.scala:417-440) + Result.fromHeap's order (:83-94): table
   * build, ADC scan, top-k and the literal TopKHeap replay of tied queries on the device.  The two `require`s stay
   * here and are re-checked natively (IllegalArgumentException either way).
   */
  def batchQueryRaw(index: Index.PQIndex, k: Int, vectors: Matrix, from: Int, until: Int): Batch = {
    require(from <= until, "expected: from <= until")
    require(from >= 0 && until <= index.length, "expected: from >= 0 && until <= length")
    val b = vectors.rows
    val kk = math.max(k, 1)
    val out = Batch(kk, new Array[Int](math.max(b, 1) * kk), new Array[Float](math.max(b, 1) * kk),
                    new Array[Int](math.max(b, 1)), new Array[Int](math.max(b, 1)))
    if (b > 0) Native.indexBatchQuery(device(index).handle, flatten(vectors.data, vectors.cols), b, k, from, until,
                                      out.idx, out.dist, out.count, out.flags)
    out
  }

  /**
   * The same as heaps, for the callers that fold them (`GroupedIndex.query` merges per-group heaps,
   * Index.scala:279).  A heap is refilled from the result in ascending order: `update` never evicts (the heap is
   * not full until the last element), so its CONTENT is exact; with equal distances among the k the reference's
   * heap may hold them in another array order than this one -- the public query path (`sortedBatchQuery`) does not
   * go through heaps and returns the reference's order bit for bit.
   */
  def batchQuery(index: Index.PQIndex, k: Int, vectors: Matrix, from: Int, until: Int): Vector[TopKHeap] = {
    val raw = batchQueryRaw(index, k, vectors, from, until)
    Vector.tabulate(vectors.rows) { q =>
      val heap = TopKHeap(k)
      var i = 0
      while (i < raw.count(q)) { heap.update(raw.idx(q * raw.k + i), raw.dist(q * raw.k + i)); i += 1 }
      heap
    }
  }

  /** SortedIndex#batchQuery (Index.scala:324-337): optional normalisation (cosine), the scan, keys looked up. */
  def sortedBatchQuery(index: Index.SortedIndex, k: Int, vectors: Matrix): Vector[Index.Result] = {
    val prepared =
      if (index.metric.normalized) Matrix(vectors.rows, vectors.cols, vectors.data.map(MathUtils.normalize(_)))
      else vectors
    val raw = batchQueryRaw(index.vectorIndex, k, prepared, 0, index.vectorIndex.length)
    Vector.tabulate(vectors.rows) { q =>
      new Index.Result(raw.rows(q).map(index.keyIndex(_)), raw.distances(q))
    }
  }

  /** Index.sorted (Index.scala:107-114): encode on the device, wrap; the device copy of the codes is made lazily. */
  def sorted(wordVectors: WordVectors.Sorted, quantizer: ProductQuantizer, metric: Metric)
            (implicit contextShift: ContextShift[IO]): IO[Index.SortedIndex] =
    HipProductQuantizer.encode(quantizer, wordVectors.toMatrix).map { encodedData =>
      Index.SortedIndex(KeyIndex.Sorted(wordVectors.keys), Index.PQIndex(quantizer, encodedData), metric)
    }

  /** Index.prepareQuery (Index.scala:352-383): B x m x k squared sub-distances, the reference's summation order. */
  def prepareQuery(pq: ProductQuantizer, queries: Array[Array[Float]]): Index.PreparedQuery = {
    val b = queries.length
    val m = pq.quantizers.size
    val k = pq.numClusters
    val flat = new Array[Float](math.max(b * m * k, 1))
    if (b > 0) Native.prepareQuery(HipProductQuantizer.flatCentroids(pq), pq.dimension, m, k, flatten(queries, pq.dimension), b, flat)
    Index.PreparedQuery(Array.tabulate(b, m)((q, j) => java.util.Arrays.copyOfRange(flat, (q * m + j) * k, (q * m + j + 1) * k)))
  }

  /** Index.exactNearestNeighbours (Index.scala:209-229): brute force over the device copy of the rows. */
  def exactNearestNeighbours(vectors: Array[Array[Float]], from: Int, until: Int, query: Array[Float], k: Int): TopKHeap = {
    require(from <= until, s"invalid range: expected from=$from <= until=$until")
    require(until <= vectors.length, s"invalid range: expected until=$until <= vectors.length=${vectors.length}")
    val heap = TopKHeap(k)
    if (until > from && k > 0) {
      val dm = DeviceMatrix.of(Matrix(vectors.length, query.length, vectors))
      val idx = new Array[Int](k); val dist = new Array[Float](k); val cnt = new Array[Int](1); val flags = new Array[Int](1)
      Native.exactKnn(dm.handle, from, until, query, 1, k, idx, dist, cnt, flags)
      var i = 0
      while (i < cnt(0)) { heap.update(idx(i), dist(i)); i += 1 }
    }
    heap
  }
}
