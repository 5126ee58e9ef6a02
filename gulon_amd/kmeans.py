"""KMeans (KMeans.scala) -- host mirror; every method body is one C-ABI call."""
import ctypes as C
from dataclasses import dataclass
from typing import Callable, List, Optional

import numpy as np

from . import native as N
from .matrix import as_device
from .vectors import Vectors

PAR_ASSIGN_BATCH = 25000   # KMeans.scala:58


@dataclass
class ProgressReport:                      # KMeans.scala:119-127
    num_iterations: int
    max_iterations: int
    step_count: int
    step_mean: np.float32
    step_s: np.float32
    converged: bool


@dataclass
class Config:                              # KMeans.scala:129-132
    num_clusters: int
    max_iterations: int
    seed: int = 0
    report: Optional[Callable[[ProgressReport], None]] = None


def _dev(vecs):
    """Vectors over a host Matrix or a DeviceMatrix -> (DeviceMatrix, from, s)."""
    dm = as_device(vecs.matrix)
    return dm, vecs.frm, vecs.dimension


class KMeans:
    def __init__(self, dimension, centroids):
        self.dimension = dimension
        self.centroids = N.f32(centroids).reshape(-1, dimension) if dimension else N.f32(centroids)

    @property
    def k(self):
        return self.centroids.shape[0]

    def __eq__(self, other):
        return (isinstance(other, KMeans) and self.dimension == other.dimension
                and np.array_equal(self.centroids, other.centroids))

    # -- KMeans.scala:18-22,70-98 (serial: one Random(0) stream over all rows)
    def assign(self, vecs: Vectors, assignments=None):
        return self._assign(vecs, 0, assignments)

    # -- KMeans.scala:57-68 (fresh Random(0) per 25 000-row batch)
    def par_assign(self, vecs: Vectors):
        return self._assign(vecs, PAR_ASSIGN_BATCH, None)

    def _assign(self, vecs, rng_batch, assignments):
        dm, frm, s = _dev(vecs)
        if assignments is None:
            assignments = np.zeros(max(dm.rows, 1), np.int32)[:dm.rows]
        out = np.ascontiguousarray(assignments, np.int32)
        N.check(N.lib().gulon_kmeans_assign(dm._h, frm, s, self.centroids.reshape(-1), self.k, rng_batch,
                                            out if out.size else np.zeros(1, np.int32)))
        return out

    # -- KMeans.scala:100-106
    def iterate(self, vecs: Vectors, iters: int):
        dm, frm, s = _dev(vecs)
        out = np.zeros_like(self.centroids)
        N.check(N.lib().gulon_kmeans_iterate(dm._h, frm, s, self.centroids.reshape(-1), self.k, iters,
                                             out.reshape(-1)))
        return KMeans(self.dimension, out)

    # -- KMeans.scala:188-196
    @staticmethod
    def init(k, vecs: Vectors, seed=0):
        dm, frm, s = _dev(vecs)
        c = np.zeros((k, s), np.float32)
        N.check(N.lib().gulon_kmeans_init(dm._h, frm, s, k, seed, c.reshape(-1), None))
        return KMeans(s, c)

    # -- KMeans.scala:198-226
    @staticmethod
    def from_assignment(k, dimension, vecs: Vectors, assignments):
        dm, frm, s = _dev(vecs)
        c = np.zeros((k, s), np.float32)
        a = N.i32(assignments)
        N.check(N.lib().gulon_kmeans_update(dm._h, frm, s, k, a if a.size else np.zeros(1, np.int32),
                                            c.reshape(-1)))
        return KMeans(dimension, c)

    # -- KMeans.scala:134-157
    @staticmethod
    def compute_clusters(vecs: Vectors, config: Config):
        dm, frm, s = _dev(vecs)
        c = np.zeros((config.num_clusters, s), np.float32)
        maxrep = config.max_iterations + 3
        reps = (N.KMeansReport * maxrep)()
        nrep = C.c_int32(0)
        N.check(N.lib().gulon_kmeans_train(dm._h, frm, s, config.num_clusters, config.max_iterations, config.seed,
                                           c.reshape(-1), reps, maxrep, C.byref(nrep)))
        if config.report is not None:
            for r in reps[:nrep.value]:
                config.report(ProgressReport(r.num_iterations, config.max_iterations, r.step_count,
                                             np.float32(r.step_mean), np.float32(r.step_s), bool(r.converged)))
        return KMeans(s, c)


def reports_to_list(reps, nrep, max_iterations) -> List[ProgressReport]:
    return [ProgressReport(r.num_iterations, max_iterations, r.step_count, np.float32(r.step_mean),
                           np.float32(r.step_s), bool(r.converged)) for r in reps[:nrep]]
