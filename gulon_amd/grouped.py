"""GroupedIndex (Index.scala:231-308) and the grouping of the vectors behind it
(WordVectors.grouped / Grouped.residuals, WordVectors.scala:24-58,118-138).

Build (CommandUtils.scala:127-147, command/BuildIndex): coarse KMeans on the full vectors ->
`group` (par_assign on the original order + stable ordering by word and cluster, group centroids, group
offsets, residual matrix on the device) -> ProductQuantizer on the residuals -> `Index.grouped`.
Row ids of a GroupedIndex are positions in the GROUPED order; `GroupedVectors.perm` maps them back.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import native as N
from .kmeans import KMeans
from .matrix import DeviceMatrix, as_device
from .product_quantizer import EncodedMatrix, ProductQuantizer
from .vectors import Vectors


@dataclass(frozen=True)
class LimitGroups:                     # GroupedIndex.Strategy.LimitGroups (Index.scala:304)
    count: int


@dataclass(frozen=True)
class LimitVectors:                    # GroupedIndex.Strategy.LimitVectors (Index.scala:305)
    count: int


@dataclass
class GroupedVectors:
    """WordVectors.Grouped without the keys (those stay with the caller)."""
    perm: np.ndarray                   # grouped position -> original row
    centroids: np.ndarray              # [g][d] centroids of the groups (a leading empty group is possible, see group)
    offsets: np.ndarray                # [g-1] first grouped row of groups 1..g-1
    residuals: DeviceMatrix            # grouped row - its group's centroid, resident in HBM

    @property
    def size(self):
        return len(self.perm)

    def cluster_of(self, i):           # WordVectors.Grouped.clusterOf (WordVectors.scala:110-113)
        return int(np.searchsorted(self.offsets, i, side="right"))


def group(vectors, clustering: KMeans, word_order=None) -> GroupedVectors:
    """WordVectors.grouped (WordVectors.scala:24-58).

    parAssign runs over the rows in the order given (the reference assigns BEFORE it sorts, so the
    25 000-row tie-break streams see the original order, :27); `word_order` = row indices stably sorted by
    word (:28-29; None: the rows are already in word order); rows are then stably ordered by cluster (:30).
    The builder loop is seeded with the cluster of ORIGINAL row 0 (`prev = assignments(0)`, :38-39): when that
    is not the lowest-numbered non-empty cluster the reference emits a leading empty group [0, 0) with a
    copy of row 0's centroid -- reproduced here (offsets[0] == 0), since it changes what LimitGroups(m)
    searches and what the index file holds."""
    if clustering.k <= 0:
        raise ValueError("requirement failed: must have at least 1 cluster")
    dm = as_device(vectors)
    assignments = clustering.par_assign(Vectors(dm))
    n = len(assignments)
    order = np.arange(n, dtype=np.int64) if word_order is None else np.asarray(word_order, np.int64)
    perm = order[np.argsort(assignments[order], kind="stable")].astype(np.int32)
    sa = assignments[perm]
    if n == 0:
        return GroupedVectors(perm, np.zeros((0, dm.cols), np.float32), np.zeros(0, np.int32), dm)
    change = np.r_[sa[0] != assignments[0], sa[1:] != sa[:-1]]          # `prev != a` at grouped position i
    offsets = np.flatnonzero(change).astype(np.int32)
    cent_ids = np.r_[assignments[0], sa[offsets]]
    centroids = np.ascontiguousarray(N.f32(clustering.centroids)[cent_ids])
    g = len(cent_ids)
    group_of = np.searchsorted(offsets, np.arange(n), side="right").astype(np.int32)
    h = C.c_void_p()
    N.check(N.lib().gulon_dataset_group_residuals(dm._h, perm, group_of, centroids.reshape(-1), g, C.byref(h)))
    return GroupedVectors(perm, centroids, offsets, DeviceMatrix(h, dm.rows, dm.cols))


class GroupedIndex:
    """Index.GroupedIndex over row ids."""

    def __init__(self, quantizer: ProductQuantizer, encoded_residuals: EncodedMatrix, centroids, offsets,
                 strategy, metric="l2"):
        self.quantizer, self.data = quantizer, encoded_residuals
        self.centroids = N.f32(centroids).reshape(-1, quantizer.dimension)
        self.offsets = np.ascontiguousarray(offsets, np.int32)
        if len(self.centroids) != len(self.offsets) + 1:        # the reference's assert, Index.scala:240-241
            raise AssertionError(f"{len(self.centroids)} != {len(self.offsets)} + 1")
        self.strategy, self.metric = strategy, metric
        h = C.c_void_p()
        packed = encoded_residuals.packed()
        N.check(N.lib().gulon_grouped_index_create(
            packed if packed.size else np.zeros(1, np.uint8), encoded_residuals.length, quantizer.dimension,
            len(quantizer.quantizers), quantizer.num_clusters, quantizer.flat_centroids(),
            self.centroids.reshape(-1), self.offsets if len(self.offsets) else np.zeros(1, np.int32),
            len(self.centroids), C.byref(h)))
        self._h = h

    @property
    def dimension(self):
        return self.quantizer.dimension

    @property
    def size(self):
        return self.data.length

    def _strategy(self):
        if isinstance(self.strategy, LimitGroups):
            return 0, int(self.strategy.count)
        if isinstance(self.strategy, LimitVectors):
            return 1, int(self.strategy.count)
        raise ValueError("strategy must be LimitGroups or LimitVectors")

    def batch_query_raw(self, k, vectors):
        from .index import normalize
        q = N.f32(vectors.data if hasattr(vectors, "data") and not isinstance(vectors, np.ndarray) else vectors)
        q = q.reshape(-1, self.dimension)
        if self.metric == "cosine" and len(q):
            q = np.stack([normalize(r) for r in q])
        b = q.shape[0]
        s, limit = self._strategy()
        oi = np.zeros((b, max(k, 1)), np.int32)
        od = np.zeros((b, max(k, 1)), np.float32)
        oc = np.zeros(max(b, 1), np.int32)
        N.check(N.lib().gulon_grouped_index_batch_query(self._h, q.reshape(-1) if b else np.zeros(1, np.float32), b, k,
                                                        s, limit, oi.reshape(-1), od.reshape(-1), oc))
        return oi[:, :k], od[:, :k], oc[:b]

    def batch_query(self, k, vectors):
        """GroupedIndex.batchQuery (Index.scala:254-257): one GroupedIndex.query per row."""
        from .index import Result
        oi, od, oc = self.batch_query_raw(k, vectors)
        return [Result(oi[i, :oc[i]].copy(), od[i, :oc[i]].copy(), 0) for i in range(len(oc))]

    def query(self, k, query):                                   # Index.scala:265-282
        return self.batch_query(k, N.f32(query).reshape(1, -1))[0]

    def close(self):
        if self._h is not None and self._h.value:
            N.lib().gulon_grouped_index_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def grouped(grouped_vectors: GroupedVectors, residuals_quantizer: ProductQuantizer, strategy, metric="l2"):
    """Index.grouped (Index.scala:133-147): encode the residuals, wrap."""
    encoded = residuals_quantizer.encode(grouped_vectors.residuals)
    return GroupedIndex(residuals_quantizer, encoded, grouped_vectors.centroids, grouped_vectors.offsets, strategy,
                        metric)
