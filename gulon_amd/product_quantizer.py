"""ProductQuantizer (ProductQuantizer.scala) and EncodedMatrix (EncodedMatrix.scala)."""
import ctypes as C
from dataclasses import dataclass
from typing import Callable, List, Optional

import numpy as np

from . import native as N
from .coder import Coder, width_for_clusters
from .kmeans import KMeans, reports_to_list
from .matrix import Matrix, as_device
from .vectors import subvector_bounds


@dataclass
class Quantizer:                            # ProductQuantizer.scala:82-86
    frm: int
    clusters: KMeans

    @property
    def dimension(self):
        return self.clusters.dimension


@dataclass
class Config:                               # ProductQuantizer.scala:107-111
    num_clusters: int
    num_quantizers: int
    max_iterations: int
    report: Optional[Callable[[List[list]], None]] = None


class EncodedMatrix:
    """coder + one packed byte array per quantizer, each covering all n rows (SoA)."""

    def __init__(self, coder: Coder, encodings):
        self.coder = coder
        self.encodings = [N.u8(e) for e in encodings]

    @property
    def length(self):
        return self.coder.length

    def indices(self):
        """[m][n] centroid ids (coder.getIndex for every row)."""
        return np.stack([self.coder.get_indices(e) for e in self.encodings]) if self.encodings else \
            np.zeros((0, self.length), np.int32)

    def packed(self):
        return np.concatenate(self.encodings) if self.encodings and self.coder.bytes_per_code else \
            np.zeros(0, np.uint8)

    def __eq__(self, other):
        return (isinstance(other, EncodedMatrix) and self.coder.width == other.coder.width
                and self.coder.length == other.coder.length
                and all(np.array_equal(a, b) for a, b in zip(self.encodings, other.encodings)))


class ProductQuantizer:
    def __init__(self, num_clusters, quantizers: List[Quantizer]):
        self.num_clusters = num_clusters
        self.quantizers = quantizers
        self.coder_width = width_for_clusters(num_clusters)          # ProductQuantizer.scala:11-16
        self.dimension = sum(q.dimension for q in quantizers)

    def coder_factory(self, length):
        return Coder(self.coder_width, length)

    def flat_centroids(self):
        """k*d floats, quantizer j's k x s_j block at k*from_j (C-ABI codebook layout)."""
        k = self.num_clusters
        out = np.zeros(k * self.dimension, np.float32)
        for q in self.quantizers:
            s = q.dimension
            out[k * q.frm: k * (q.frm + s)] = q.clusters.centroids.reshape(-1)
        return out

    @staticmethod
    def from_flat(num_clusters, d, m, cents):
        fr, un = subvector_bounds(d, m)
        k = num_clusters
        qs = [Quantizer(int(f), KMeans(int(u - f), N.f32(cents[k * f: k * u]).reshape(k, u - f)))
              for f, u in zip(fr, un)]
        return ProductQuantizer(num_clusters, qs)

    # -- ProductQuantizer.scala:150-153 / :121-148
    @staticmethod
    def apply(vectors, config: Config):
        dm = as_device(vectors)
        m, k = config.num_quantizers, config.num_clusters
        cents = np.zeros(k * dm.cols, np.float32)
        maxrep = config.max_iterations + 3
        reps = (N.KMeansReport * (m * maxrep))()
        nrep = (C.c_int32 * m)()
        N.check(N.lib().gulon_pq_train(dm._h, m, k, config.max_iterations, cents, reps, maxrep,
                                       C.cast(nrep, C.c_void_p)))
        if config.report is not None:
            config.report([reports_to_list(reps[j * maxrep:(j + 1) * maxrep], nrep[j], config.max_iterations)
                           for j in range(m)])
        return ProductQuantizer.from_flat(k, dm.cols, m, cents)

    # -- ProductQuantizer.scala:25-35
    def encode(self, vectors) -> EncodedMatrix:
        dm = as_device(vectors)
        m = len(self.quantizers)
        coder = self.coder_factory(dm.rows)
        buf = np.zeros(max(m * coder.bytes_per_code, 1), np.uint8)
        N.check(N.lib().gulon_pq_encode(dm._h, m, self.num_clusters, self.flat_centroids(), buf))
        b = coder.bytes_per_code
        return EncodedMatrix(coder, [buf[j * b:(j + 1) * b].copy() for j in range(m)])

    # -- ProductQuantizer.scala:37-78 (host-side table lookup)
    def decode(self, encoded: EncodedMatrix) -> Matrix:
        idx = encoded.indices()
        out = np.zeros((encoded.length, self.dimension), np.float32)
        for j, q in enumerate(self.quantizers):
            out[:, q.frm:q.frm + q.dimension] = q.clusters.centroids[idx[j]]
        return Matrix(out)
