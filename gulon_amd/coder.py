"""Coder (Coder.scala): bit-packing of centroid ids, widths 0/2/4/8/10/12/16."""
import ctypes as C

import numpy as np

from . import native as N


def width_for_clusters(num_clusters):
    """ProductQuantizer.coderFactory (ProductQuantizer.scala:11-16)."""
    w = C.c_int32(0)
    rc = N.lib().gulon_coder_width(num_clusters, C.byref(w))
    if rc != N.OK:
        raise ValueError(f"too many clusters: {num_clusters}")
    return w.value


class Coder:
    def __init__(self, width, length):
        nb = C.c_int32(0)
        rc = N.lib().gulon_coder_bytes(width, length, C.byref(nb))
        if rc != N.OK:
            raise ValueError(f"unsupported width: {width}")       # Coder.scala:57
        self.width, self.length, self.bytes_per_code = width, length, nb.value

    def build_code(self, indices):
        idx = N.i32(indices)
        if idx.size != self.length:
            raise ValueError(f"indices.length != {self.length}")   # Coder.scala:148-150
        code = np.zeros(max(self.bytes_per_code, 1), np.uint8)
        N.check(N.lib().gulon_coder_build(self.width, idx if idx.size else np.zeros(1, np.int32), self.length, code))
        return code[:self.bytes_per_code]

    def get_indices(self, code):
        out = np.zeros(max(self.length, 1), np.int32)
        c = N.u8(code)
        if c.size == 0:
            c = np.zeros(1, np.uint8)
        N.check(N.lib().gulon_coder_unpack(self.width, c, self.length, out))
        return out[:self.length]

    def get_index(self, code, i):
        if i < 0 or i >= self.length:
            raise IndexError(i)
        return int(self.get_indices(code)[i])
