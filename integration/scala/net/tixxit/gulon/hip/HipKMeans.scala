package net.tixxit.gulon.hip

import cats.effect.{ContextShift, IO}
import cats.implicits._

import net.tixxit.gulon.{KMeans, SummaryStats, Vectors}

/**
 * Replacement bodies of the KMeans hot path (KMeans.scala).  The public signatures stay in
 * `KMeans` / `object KMeans`; each delegates here with one line, e.g.
 * {{{
 *   // class KMeans
 *   def assign(vecs: Vectors): Array[Int]                    = hip.HipKMeans.assign(this, vecs)       // :18-22
 *   def assign(vecs: Vectors, assignments: Array[Int]): Unit = hip.HipKMeans.assign(this, vecs, assignments) // :70-98
 *   def parAssign(vecs: Vectors)(implicit cs: ContextShift[IO]): IO[Array[Int]] = hip.HipKMeans.parAssign(this, vecs) // :57-68
 *   def iterate(vecs: Vectors, iters: Int): KMeans           = hip.HipKMeans.iterate(this, vecs, iters) // :100-106
 *   // object KMeans
 *   def computeClusters(vecs: Vectors, config: Config)(implicit cs: ContextShift[IO]): IO[KMeans] =
 *     hip.HipKMeans.computeClusters(vecs, config)                                                      // :134-157
 *   def init(k: Int, vecs: Vectors, seed: Int = 0): KMeans   = hip.HipKMeans.init(k, vecs, seed)       // :188-196
 *   def fromAssignment(k: Int, dimension: Int, vecs: Vectors, assignments: Array[Int]): KMeans =
 *     hip.HipKMeans.fromAssignment(k, dimension, vecs, assignments)                                    // :198-226
 * }}}
 * `KMeans.apply(dimension, centroids)` (:170-186, the offsets) stays as it is: the native side
 * recomputes the offsets with the same sequential sum from the centroids it is handed.
 * NOT compiled in this repository (no JVM toolchain in the build image).
 */
object HipKMeans {
  /** KMeans.scala:58 -- parAssign restarts `new Random(0)` every 25 000 rows; the serial forms never do. */
  final val ParAssignBatch = 25000
  final val SerialStream = 0

  private def flat(centroids: Array[Array[Float]], s: Int): Array[Float] = {
    val out = new Array[Float](centroids.length * s)
    var i = 0
    while (i < centroids.length) { System.arraycopy(centroids(i), 0, out, i * s, s); i += 1 }
    out
  }

  private def unflat(flat: Array[Float], k: Int, s: Int): Array[Array[Float]] =
    Array.tabulate(k)(i => java.util.Arrays.copyOfRange(flat, i * s, (i + 1) * s))

  /** KMeans#assign(vecs) (KMeans.scala:18-22) and #assign(vecs, assignments) (:70-98): one Random(0) stream. */
  def assign(self: KMeans, vecs: Vectors): Array[Int] = {
    val out = new Array[Int](vecs.size)
    assign(self, vecs, out)
    out
  }

  def assign(self: KMeans, vecs: Vectors, assignments: Array[Int]): Unit = {
    // a row whose distances are all NaN keeps the slot's previous content (KMeans.scala:86-89), which is
    // why the caller's array goes in AND out
    val dm = DeviceMatrix.of(vecs.matrix)
    Native.kmeansAssign(dm.handle, vecs.from, vecs.dimension, flat(self.centroids, vecs.dimension), self.k,
                        SerialStream, assignments)
  }

  /** KMeans#parAssign (KMeans.scala:57-68): same arithmetic, tie-break stream restarted per 25 000-row batch. */
  def parAssign(self: KMeans, vecs: Vectors)(implicit contextShift: ContextShift[IO]): IO[Array[Int]] =
    IO.shift *> IO.delay {
      val out = new Array[Int](vecs.size)
      val dm = DeviceMatrix.of(vecs.matrix)
      Native.kmeansAssign(dm.handle, vecs.from, vecs.dimension, flat(self.centroids, vecs.dimension), self.k,
                          ParAssignBatch, out)
      out
    }

  /** KMeans#iterate (KMeans.scala:100-106). */
  def iterate(self: KMeans, vecs: Vectors, iters: Int): KMeans = {
    val s = vecs.dimension
    val out = new Array[Float](self.k * s)
    Native.kmeansIterate(DeviceMatrix.of(vecs.matrix).handle, vecs.from, s, flat(self.centroids, s), self.k, iters, out)
    if (iters <= 0) self else KMeans(self.dimension, unflat(out, self.k, s))
  }

  /** KMeans.init (KMeans.scala:188-196): k draws of java.util.Random(seed).nextInt(n), with replacement. */
  def init(k: Int, vecs: Vectors, seed: Int): KMeans = {
    val s = vecs.dimension
    val out = new Array[Float](k * s)
    Native.kmeansInit(DeviceMatrix.of(vecs.matrix).handle, vecs.from, s, k, seed, out)
    KMeans(s, unflat(out, k, s))
  }

  /** KMeans.fromAssignment (KMeans.scala:198-226): the order-dependent running mean, bit for bit. */
  def fromAssignment(k: Int, dimension: Int, vecs: Vectors, assignments: Array[Int]): KMeans = {
    val s = vecs.dimension
    val out = new Array[Float](k * s)
    Native.kmeansUpdate(DeviceMatrix.of(vecs.matrix).handle, vecs.from, s, k, assignments, out)
    KMeans(dimension, unflat(out, k, s))
  }

  /** The ProgressReports of one training run, as the native side returned them (KMeans.scala:119-127). */
  private[hip] def reports(maxIterations: Int, ints: Array[Int], floats: Array[Float], at: Int,
                           n: Int): Vector[KMeans.ProgressReport] =
    Vector.tabulate(n) { r =>
      val i = at + r
      KMeans.ProgressReport(numIterations = ints(3 * i), maxIterations = maxIterations,
                            stepSize = SummaryStats(ints(3 * i + 2), floats(2 * i), floats(2 * i + 1)),
                            converged = ints(3 * i + 1) != 0)
    }

  /** Reports per run: the one after init (:141-142) + one per executed iteration i = 0 .. maxIterations (:144-153). */
  private[hip] def maxReports(maxIterations: Int): Int = math.max(maxIterations, 0) + 3

  /**
   * KMeans.computeClusters (KMeans.scala:134-157).  The whole loop -- init, parAssign, fromAssignment,
   * Arrays.equals, stepSize -- runs on the device in one blocking call; `config.report` is then replayed
   * with exactly the reports the reference would have produced, in order, before the IO completes.
   */
  def computeClusters(vecs: Vectors, config: KMeans.Config)(implicit contextShift: ContextShift[IO]): IO[KMeans] =
    for {
      _ <- IO.shift
      trained <- IO.delay {
        val s = vecs.dimension
        val cents = new Array[Float](config.numClusters * s)
        val mr = maxReports(config.maxIterations)
        val ints = new Array[Int](3 * mr)
        val floats = new Array[Float](2 * mr)
        val n = new Array[Int](1)
        Native.kmeansTrain(DeviceMatrix.of(vecs.matrix).handle, vecs.from, s, config.numClusters,
                           config.maxIterations, config.seed, cents, ints, floats, mr, n)
        (KMeans(s, unflat(cents, config.numClusters, s)), reports(config.maxIterations, ints, floats, 0, math.min(n(0), mr)))
      }
      _ <- trained._2.traverse_(config.report)
    } yield trained._1
}
