"""Matrix (Matrix.scala:3) and its device-resident form.

The reference's Matrix is a jagged Array[Array[Float]]; across the C ABI it is one
flat row-major float32 buffer (ld = cols).  DeviceMatrix owns the HBM copy that
KMeans / ProductQuantizer / exact kNN read.
"""
import ctypes as C

import numpy as np

from . import native as N


class Matrix:
    def __init__(self, data):
        self.data = N.f32(data)
        if self.data.ndim != 2:
            raise ValueError("Matrix data must be 2-D")

    @property
    def rows(self):
        return self.data.shape[0]

    @property
    def cols(self):
        return self.data.shape[1]

    def __eq__(self, other):
        return isinstance(other, Matrix) and np.array_equal(self.data, other.data)


class DeviceMatrix:
    """n x d float32 matrix resident in HBM (gulon_dataset)."""

    def __init__(self, handle, n, d):
        self._h = handle
        self.rows, self.cols = n, d

    @classmethod
    def from_host(cls, data):
        a = N.f32(data)
        if a.ndim != 2 or a.shape[1] < 1:
            raise ValueError("need an n x d array with d >= 1")
        h = C.c_void_p()
        N.check(N.lib().gulon_dataset_create(a.reshape(-1) if a.size else np.zeros(1, np.float32),
                                             a.shape[0], a.shape[1], C.byref(h)))
        return cls(h, a.shape[0], a.shape[1])

    @classmethod
    def synthetic(cls, n, d, kind, seed, ncentres=1000):
        """kind: 0 iid N(0,1), 1 clustered, 2 U[0,1), 3 overlapping clusters -- bit-identical to oracle.synth."""
        h = C.c_void_p()
        N.check(N.lib().gulon_dataset_create_synth(n, d, kind, seed, ncentres, C.byref(h)))
        return cls(h, n, d)

    def device_ptr(self):
        p = C.c_void_p()
        N.check(N.lib().gulon_dataset_device_ptr(self._h, C.byref(p)))
        return p.value

    def get_rows(self, rows):
        rows = N.i32(rows)
        out = np.zeros((rows.size, self.cols), np.float32)
        if rows.size:
            N.check(N.lib().gulon_dataset_get_rows(self._h, rows, rows.size, out.reshape(-1)))
        return out

    def to_host(self, chunk=1 << 20):
        out = np.zeros((self.rows, self.cols), np.float32)
        for s in range(0, self.rows, chunk):
            e = min(self.rows, s + chunk)
            out[s:e] = self.get_rows(np.arange(s, e, dtype=np.int32))
        return out

    def close(self):
        if self._h is not None and self._h.value:
            N.lib().gulon_dataset_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def as_device(m):
    if isinstance(m, DeviceMatrix):
        return m
    if isinstance(m, Matrix):
        return DeviceMatrix.from_host(m.data)
    return DeviceMatrix.from_host(m)
