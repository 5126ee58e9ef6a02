package net.tixxit.gulon.hip

import java.util.{Collections, WeakHashMap}

import net.tixxit.gulon.Matrix

/**
 * A `Matrix` (Matrix.scala:3) resident in HBM: `gulon_dataset` behind a handle.
 *
 * `Matrix` is an immutable case class whose `hashCode` / `equals` walk every element
 * (Matrix.scala:4-10), so the cache is keyed on the identity of its `data` array (arrays hash by
 * identity): every `Vectors` view of one `Matrix` (Vectors.scala:3: `data = matrix.data`) and
 * every call of the training loop then reuse ONE upload.  Entries die with the array (weak keys);
 * `release` frees the device copy early.
 */
final class DeviceMatrix private (val handle: Long, val rows: Int, val cols: Int) {
  @volatile private var open = true
  def close(): Unit = synchronized { if (open) { open = false; Native.datasetDestroy(handle) } }
  override def finalize(): Unit = close()
}

object DeviceMatrix {
  private val cache = Collections.synchronizedMap(new WeakHashMap[Array[Array[Float]], DeviceMatrix]())

  /** The device copy of `m`, uploaded on first use (one direct row-major buffer, ld = cols). */
  def of(m: Matrix): DeviceMatrix = cache.synchronized {
    val hit = cache.get(m.data)
    if (hit != null) hit
    else {
      val h = Native.datasetCreate(Native.flatten(m.data, m.cols), m.rows, m.cols)
      val dm = new DeviceMatrix(h, m.rows, m.cols)
      cache.put(m.data, dm)
      dm
    }
  }

  /** A matrix that already lives on the device (`Native.groupResiduals`). */
  def adopt(handle: Long, rows: Int, cols: Int): DeviceMatrix = new DeviceMatrix(handle, rows, cols)

  def release(m: Matrix): Unit = {
    val dm = cache.remove(m.data)
    if (dm != null) dm.close()
  }
}
