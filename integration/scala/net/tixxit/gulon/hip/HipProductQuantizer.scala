package net.tixxit.gulon.hip

import cats.effect.{ContextShift, IO}
import cats.implicits._

import net.tixxit.gulon.{EncodedMatrix, KMeans, Matrix, ProductQuantizer, Vectors}

/**
 * Replacement bodies of ProductQuantizer.apply / fromSubvectors (ProductQuantizer.scala:121-153) and
 * ProductQuantizer#encode (:25-35).  The reference keeps its signatures and delegates:
 * {{{
 *   // case class ProductQuantizer
 *   def encode(vectors: Matrix)(implicit cs: ContextShift[IO]): IO[EncodedMatrix] = hip.HipProductQuantizer.encode(this, vectors)
 *   // object ProductQuantizer
 *   def apply(vectors: Matrix, config: Config)(implicit cs: ContextShift[IO]): IO[ProductQuantizer] =
 *     hip.HipProductQuantizer(vectors, config)
 * }}}
 * NOT compiled in this repository (no JVM toolchain in the build image).
 */
object HipProductQuantizer {
  /** The C ABI's codebook layout: k*d floats, quantizer j's k x s_j block at k * from_j (include/gulon_hip.h). */
  def flatCentroids(pq: ProductQuantizer): Array[Float] = {
    val k = pq.numClusters
    val out = new Array[Float](k * pq.dimension)
    pq.quantizers.foreach { q =>
      val s = q.dimension
      var c = 0
      while (c < k) { System.arraycopy(q.clusters.centroids(c), 0, out, k * q.from + c * s, s); c += 1 }
    }
    out
  }

  def fromFlat(numClusters: Int, subvectors: Seq[Vectors], cents: Array[Float]): ProductQuantizer =
    ProductQuantizer(numClusters, subvectors.iterator.map { v =>
      val s = v.dimension
      val base = numClusters * v.from
      val cs = Array.tabulate(numClusters)(c => java.util.Arrays.copyOfRange(cents, base + c * s, base + (c + 1) * s))
      ProductQuantizer.Quantizer(v.from, KMeans(s, cs))
    }.toVector)

  /**
   * ProductQuantizer.apply (ProductQuantizer.scala:150-153).  `Vectors.subvectors` (Vectors.scala:84-104) stays in
   * Scala; the m independent computeClusters (seed = quantizer index, :139) run iteration-synchronously on the device.
   * The reference reports from m concurrent fibres, each `makeReport` updating slot i of a shared vector and
   * handing the whole vector to `config.report` (:124-136): any interleaving of the per-quantizer sequences is a
   * schedule the reference can produce.  Replayed here round by round -- report r of quantizer 0, 1, ..., m-1,
   * then report r + 1 -- which is the schedule of a pool that runs the fibres in lockstep.
   */
  def apply(vectors: Matrix, config: ProductQuantizer.Config)(implicit contextShift: ContextShift[IO]): IO[ProductQuantizer] = {
    val subvectors = Vectors.subvectors(vectors, config.numQuantizers).toVector
    val m = subvectors.size
    for {
      _ <- IO.shift
      trained <- IO.delay {
        val cents = new Array[Float](config.numClusters * vectors.cols)
        val mr = HipKMeans.maxReports(config.maxIterations)
        val ints = new Array[Int](3 * m * mr)
        val floats = new Array[Float](2 * m * mr)
        val counts = new Array[Int](m)
        Native.pqTrain(DeviceMatrix.of(vectors).handle, m, config.numClusters, config.maxIterations, cents, ints, floats,
                       mr, counts)
        val perQuantizer = Vector.tabulate(m)(j =>
          HipKMeans.reports(config.maxIterations, ints, floats, j * mr, math.min(counts(j), mr)))
        (fromFlat(config.numClusters, subvectors, cents), perQuantizer)
      }
      _ <- {
        // (no tuple pattern in the for: IO has no withFilter)
        val perQuantizer = trained._2
        val rounds = if (perQuantizer.isEmpty) 0 else perQuantizer.map(_.size).max
        val steps = for { r <- 0 until rounds; j <- 0 until m if r < perQuantizer(j).size } yield (j, perQuantizer(j)(r))
        val snapshots = steps.scanLeft(Vector.fill(m)(KMeans.ProgressReport.init(config.maxIterations))) {
          case (current, (j, report)) => current.updated(j, report)
        }.tail
        snapshots.toList.traverse_(rs => config.report(ProductQuantizer.ProgressReport(rs)))
      }
    } yield trained._1
  }

  /**
   * ProductQuantizer#encode (ProductQuantizer.scala:25-35): per quantizer the SERIAL assign (one Random(0) stream
   * over all rows, KMeans.scala:70-98), packed by the Coder for numClusters (:11-16) on the device; the m packed
   * arrays come back to back and are wrapped by the reference's own `coder.wrapCode`.
   */
  def encode(self: ProductQuantizer, vectors: Matrix)(implicit contextShift: ContextShift[IO]): IO[EncodedMatrix] =
    IO.shift *> IO.delay {
      val coder = self.coderFactory(vectors.rows)
      val m = self.quantizers.size
      val bytesPerCode = coder.unwrapCode(coder.buildCode(new Array[Int](vectors.rows))).length
      val packed = new Array[Byte](math.max(m * bytesPerCode, 1))
      Native.pqEncode(DeviceMatrix.of(vectors).handle, m, self.numClusters, flatCentroids(self), packed)
      val codes = Vector.tabulate(m) { j =>
        coder.wrapCode(java.util.Arrays.copyOfRange(packed, j * bytesPerCode, (j + 1) * bytesPerCode))
      }
      EncodedMatrix(coder)(codes)
    }
}
