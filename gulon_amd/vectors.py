"""Vectors (Vectors.scala): a (matrix, from, until) column-slice view and the
subvector split rule (Vectors.scala:84-104)."""
import numpy as np

from . import native as N


class Vectors:
    def __init__(self, matrix, frm=0, until=None):
        self.matrix = matrix
        self.frm = frm
        self.until = matrix.cols if until is None else until

    @property
    def dimension(self):
        return self.until - self.frm

    @property
    def size(self):
        return self.matrix.rows


def subvector_bounds(d, m):
    fr = np.zeros(m, np.int32)
    un = np.zeros(m, np.int32)
    N.check(N.lib().gulon_subvectors(d, m, fr, un))
    return fr, un


def subvectors(matrix, num_subvectors):
    fr, un = subvector_bounds(matrix.cols, num_subvectors)
    return [Vectors(matrix, int(f), int(u)) for f, u in zip(fr, un)]
