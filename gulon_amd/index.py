"""Index (Index.scala): prepareQuery, PQIndex, SortedIndex, exactNearestNeighbours."""
import ctypes as C
import os
import weakref
from dataclasses import dataclass

import numpy as np

from . import native as N
from .matrix import Matrix, as_device
from .product_quantizer import EncodedMatrix, ProductQuantizer


@dataclass
class Result:
    """Index.Result (Index.scala:56-74) with int row ids instead of String keys
    (KeyIndex stays on the JVM side): ascending squared-L2 distances."""
    rows: np.ndarray
    distances: np.ndarray
    flags: int = 0

    def __len__(self):
        return len(self.rows)

    def __iter__(self):
        return iter(zip(self.rows.tolist(), self.distances.tolist()))


def prepare_query(pq: ProductQuantizer, queries):
    """Index.prepareQuery (Index.scala:352-383) -> [B][m][k] float32."""
    q = N.f32(queries)
    b, d = q.shape
    m, k = len(pq.quantizers), pq.num_clusters
    t = np.zeros((b, m, k), np.float32)
    N.check(N.lib().gulon_prepare_query(pq.flat_centroids(), d, m, k, q.reshape(-1) if b else np.zeros(1, np.float32),
                                        b, t.reshape(-1) if t.size else np.zeros(1, np.float32)))
    return t


def exact_nearest_neighbours(vectors, query, k, frm=0, until=None):
    """Index.exactNearestNeighbours (Index.scala:209-229) for one or many queries."""
    dm = as_device(vectors)
    q = N.f32(query)
    single = q.ndim == 1
    q = q.reshape(1, -1) if single else q
    until = dm.rows if until is None else until
    b = q.shape[0]
    oi = np.zeros((b, max(k, 1)), np.int32)
    od = np.zeros((b, max(k, 1)), np.float32)
    oc = np.zeros(max(b, 1), np.int32)
    of = np.zeros(max(b, 1), np.int32)
    N.check(N.lib().gulon_exact_knn(dm._h, frm, until, q.reshape(-1), b, k, oi.reshape(-1), od.reshape(-1), oc, of))
    res = [Result(oi[i, :oc[i]].copy(), od[i, :oc[i]].copy(), int(of[i])) for i in range(b)]
    return res[0] if single else res


_LIVE = weakref.WeakSet()     # open PQIndex handles and contexts (tune_live)


def tune_live(**knobs):
    """Tuning experiments and tests: sets the launch-shape knobs (GULON_SCAN_FILTER=0, GULON_FILTER_CAP=64, ...) in the
    environment -- where every handle created from now on takes them from -- and on every open handle and context
    (gulon_index_tuning).  The library has no process-wide setter; results never depend on the knobs."""
    for key, value in knobs.items():
        os.environ[key] = str(int(value))
        for ix in list(_LIVE):
            if ix._h is not None and ix._h.value:
                N.check(N.lib().gulon_index_tuning(ix._h, key.encode(), int(value)))


class PQIndex:
    """PQIndex(productQuantizer, data) (Index.scala:385-441): owns the HBM copy of the codes."""

    def __init__(self, product_quantizer: ProductQuantizer, data: EncodedMatrix, row_base=0, _handle=None):
        self.product_quantizer = product_quantizer
        self.data = data
        self.row_base = row_base
        _LIVE.add(self)
        if _handle is not None:
            self._h = _handle
            return
        h = C.c_void_p()
        packed = data.packed()
        N.check(N.lib().gulon_index_create(packed if packed.size else np.zeros(1, np.uint8), data.length,
                                           product_quantizer.dimension, len(product_quantizer.quantizers),
                                           product_quantizer.num_clusters, product_quantizer.flat_centroids(),
                                           row_base, C.byref(h)))
        self._h = h

    def context(self):
        """Another workspace over the same device-resident codes and codebooks (gulon_index_context_create):
        one per batch in flight / per querying thread; the codes live until the index and all its contexts
        are closed."""
        h = C.c_void_p()
        N.check(N.lib().gulon_index_context_create(self._h, C.byref(h)))
        return PQIndex(self.product_quantizer, self.data, self.row_base, _handle=h)

    @property
    def dimension(self):
        return self.product_quantizer.dimension

    @property
    def length(self):
        return self.data.length

    def batch_query_raw(self, k, vectors, frm=0, until=None):
        q = vectors.data if isinstance(vectors, Matrix) else N.f32(vectors)
        q = N.f32(q).reshape(-1, self.dimension)
        until = self.length if until is None else until
        b = q.shape[0]
        oi = np.zeros((b, max(k, 1)), np.int32)
        od = np.zeros((b, max(k, 1)), np.float32)
        oc = np.zeros(max(b, 1), np.int32)
        of = np.zeros(max(b, 1), np.int32)
        N.check(N.lib().gulon_index_batch_query(self._h, q.reshape(-1) if b else np.zeros(1, np.float32), b, k, frm,
                                                until, oi.reshape(-1), od.reshape(-1), oc, of))
        return oi[:, :k], od[:, :k], oc[:b], of[:b]

    def batch_query(self, k, vectors, frm=0, until=None):
        """PQIndex.batchQuery (Index.scala:417-440) + Result.fromHeap (Index.scala:83-94)."""
        oi, od, oc, of = self.batch_query_raw(k, vectors, frm, until)
        return [Result(oi[i, :oc[i]].copy(), od[i, :oc[i]].copy(), int(of[i])) for i in range(len(oc))]

    def query(self, k, query, frm=0, until=None):                     # Index.scala:411-412
        return self.batch_query(k, N.f32(query).reshape(1, -1), frm, until)[0]

    def decode(self, row):                                            # Index.scala:390-391
        idx = self.data.indices()[:, row]
        out = np.zeros(self.dimension, np.float32)
        for j, q in enumerate(self.product_quantizer.quantizers):
            out[q.frm:q.frm + q.dimension] = q.clusters.centroids[idx[j]]
        return out

    def close(self):
        if self._h is not None and self._h.value:
            N.lib().gulon_index_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def normalize(xs):
    """MathUtils.normalize (MathUtils.scala:100-120): sequential fp32 sum, math.sqrt in double."""
    xs = N.f32(xs)
    s = np.float32(0)
    for x in xs:
        s = np.float32(s + np.float32(x * x))
    dist = np.float32(np.sqrt(np.float64(s)))
    return (xs / dist).astype(np.float32)


class SortedIndex:
    """Index.SortedIndex (Index.scala:310-337) without the String key index."""

    def __init__(self, vector_index: PQIndex, metric="l2"):
        self.vector_index = vector_index
        self.metric = metric

    @property
    def dimension(self):
        return self.vector_index.dimension

    @property
    def size(self):
        return self.vector_index.length

    def _prepare(self, q):                                            # Index.scala:324-331
        q = N.f32(q.data if isinstance(q, Matrix) else q).reshape(-1, self.dimension)
        if self.metric == "cosine":
            q = np.stack([normalize(r) for r in q]) if len(q) else q
        return q

    def batch_query(self, k, vectors):                                # Index.scala:333-336
        return self.vector_index.batch_query(k, self._prepare(vectors))

    def query(self, k, vector):                                       # Index.scala:321-322
        return self.batch_query(k, N.f32(vector).reshape(1, -1))[0]

    def lookup_row(self, row):                                        # Index.scala:318-319
        return self.vector_index.decode(row)


class Index:
    @staticmethod
    def sorted(vectors, quantizer: ProductQuantizer, metric="l2") -> SortedIndex:
        """Index.sorted (Index.scala:107-114): encode, then wrap."""
        encoded = quantizer.encode(as_device(vectors))
        return SortedIndex(PQIndex(quantizer, encoded), metric)

    @staticmethod
    def grouped(grouped_vectors, residuals_quantizer: ProductQuantizer, strategy, metric="l2"):
        """Index.grouped (Index.scala:133-147): the quantizer is one on the RESIDUALS."""
        from .grouped import grouped
        return grouped(grouped_vectors, residuals_quantizer, strategy, metric)
