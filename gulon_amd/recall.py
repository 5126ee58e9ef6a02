"""Recall harness (Tests.scala): queries sampled from the dataset with
java.util.Random(seed), exact kNN ground truth, distance-threshold recall."""
import numpy as np

from . import native as N
from .index import exact_nearest_neighbours
from .matrix import as_device


class _JRandom:
    """java.util.Random restated for host-side sampling (Tests.scala:82-84)."""
    _M = (1 << 48) - 1

    def __init__(self, seed):
        self.s = (seed ^ 0x5DEECE66D) & self._M

    def _next(self, bits):
        self.s = (self.s * 0x5DEECE66D + 0xB) & self._M
        v = (self.s >> (48 - bits)) & 0xFFFFFFFF
        return v - (1 << 32) if v >= (1 << 31) else v

    def next_int(self, bound):
        r = self._next(31)
        m = bound - 1
        if bound & m == 0:
            return (bound * r) >> 31
        u = r
        while True:
            r = u % bound
            if ((u - r + m) & 0xFFFFFFFF) < (1 << 31):
                return r
            u = self._next(31)


def sample_rows(n, sample_size, seed=0):
    rng = _JRandom(seed)
    return np.array([rng.next_int(n) for _ in range(sample_size)], np.int32)


def recall_at_k(vectors, queries, k, ann_rows, ann_count):
    """Tests.recallOf (Tests.scala:18-41), eps = 0, one k: mean and stdDev of
    (#returned rows whose exact distanceSq <= exact K-th distance) / k."""
    dm = as_device(vectors)
    q = N.f32(queries)
    b = q.shape[0]
    truth = exact_nearest_neighbours(dm, q, k)
    rows = np.full((b, k), -1, np.int32)
    for i in range(b):
        c = min(int(ann_count[i]), k)
        rows[i, :c] = ann_rows[i][:c]
    dist = np.zeros((b, k), np.float32)
    N.check(N.lib().gulon_distance_sq_rows(dm._h, q.reshape(-1), b, rows.reshape(-1), k, dist.reshape(-1)))
    cnt, mean, ss = 0, np.float32(0), np.float32(0)
    for i in range(b):
        if len(truth[i]) < k:
            continue
        cutoff = truth[i].distances[k - 1]
        tp = int(np.sum((dist[i] <= cutoff) & (rows[i] >= 0)))
        x = np.float32(tp) / np.float32(k)
        if cnt == 0:
            cnt, mean, ss = 1, x, np.float32(0)
        else:                                   # SummaryStats.++ (MathUtils.scala:11-22)
            n = cnt + 1
            dlt = np.float32(mean - x)
            nm = np.float32(mean + np.float32(np.float32(1) / np.float32(n)) * np.float32(x - mean))
            ns = np.float32(np.float32(ss + np.float32(0)) +
                            np.float32(np.float32(np.float32(np.float32(dlt * dlt) * np.float32(cnt)) * np.float32(1))
                                       / np.float32(n)))
            cnt, mean, ss = n, nm, ns
    sd = float(np.sqrt(np.float64(ss / np.float32(cnt)))) if cnt else 0.0
    return float(mean), sd
