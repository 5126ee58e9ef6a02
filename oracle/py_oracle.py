"""Second, independent restatement of Gulon's hot path in pure Python with
numpy.float32 scalars, written from the Scala source (not from the C oracle).

TEST INFRASTRUCTURE ONLY; small inputs only (pure-Python loops).  Its single
purpose is to cross-check oracle/gulon_oracle.c bit for bit.  PARITY UNPINNED
vs the JVM (see gulon_oracle.c header).  Citations are relative to
/root/reference/core/src/main/scala/net/tixxit/gulon/.
"""
import numpy as np

f32 = np.float32
_MASK = (1 << 48) - 1
_FLT_MAX = np.finfo(np.float32).max


class JRandom:
    """java.util.Random per the JDK specification."""

    def __init__(self, seed):
        self.seed = (int(seed) ^ 0x5DEECE66D) & _MASK

    def _next(self, bits):
        self.seed = (self.seed * 0x5DEECE66D + 0xB) & _MASK
        v = self.seed >> (48 - bits)
        v &= 0xFFFFFFFF
        return v - (1 << 32) if v >= (1 << 31) else v

    def next_int(self, bound=None):
        if bound is None:
            return self._next(32)
        r = self._next(31)
        m = bound - 1
        if bound & m == 0:
            return (bound * r) >> 31
        u = r
        while True:
            r = u % bound
            t = (u - r + m) & 0xFFFFFFFF
            if t < (1 << 31):
                return r
            u = self._next(31)

    def next_boolean(self):
        return self._next(1) != 0


def subvectors(d, m):                                   # Vectors.scala:84-104
    ideal = (d + m - 1) // m
    short = ideal * m - d
    full = m - short
    out = []
    for i in range(m):
        if i < full:
            out.append((i * ideal, i * ideal + ideal))
        else:
            fr = full * ideal + (i - full) * (ideal - 1)
            out.append((fr, fr + ideal - 1))
    return out


def distance_sq(x, y):                                  # MathUtils.scala:85-95
    s = f32(0)
    for a, b in zip(x, y):
        dx = f32(b) - f32(a)
        s = f32(s + f32(dx * dx))
    return s


class Heap:                                             # TopKHeap.scala
    def __init__(self, k):
        self.keys = [0] * k
        self.values = [f32(0)] * k
        self.size = 0

    def _swap(self, i, j):
        self.keys[i], self.keys[j] = self.keys[j], self.keys[i]
        self.values[i], self.values[j] = self.values[j], self.values[i]

    def _up(self, i):
        if i > 0:
            p = (i - 1) // 2
            if self.values[i] > self.values[p]:
                self._swap(i, p)
                self._up(p)

    def _down(self, i):
        top, lc, rc = i, 2 * i + 1, 2 * i + 2
        if lc < self.size and self.values[top] < self.values[lc]:
            top = lc
        if rc < self.size and self.values[top] < self.values[rc]:
            top = rc
        if top != i:
            self._swap(i, top)
            self._down(top)

    def delete(self):
        if self.size <= 0:
            raise RuntimeError("heap is empty")
        self.size -= 1
        removed = self.keys[0]
        self.keys[0] = self.keys[self.size]
        self.values[0] = self.values[self.size]
        self._down(0)
        return removed

    def update(self, k, v):
        v = f32(v)
        if self.size == len(self.keys) and self.size > 0 and self.values[0] > v:
            self.delete()
        if self.size < len(self.keys):
            self.keys[self.size] = k
            self.values[self.size] = v
            self._up(self.size)
            self.size += 1

    def merge(self, that):
        for i in range(that.size):
            self.update(that.keys[i], that.values[i])

    def drain(self):                                     # Index.scala:83-94
        n = self.size
        ks, vs = [0] * n, [f32(0)] * n
        for i in range(n - 1, -1, -1):
            ks[i], vs[i] = self.keys[0], self.values[0]
            self.delete()
        return ks, vs


def kmeans_offsets(cents):                              # KMeans.scala:170-186
    out = []
    for c in cents:
        s = f32(0)
        for x in c:
            s = f32(s + f32(f32(x) * f32(x)))
        out.append(s)
    return out


def kmeans_init(X, fr, un, k, seed=0):                  # KMeans.scala:188-196
    rng = JRandom(seed)
    return [[f32(v) for v in X[rng.next_int(len(X))][fr:un]] for _ in range(k)]


def kmeans_assign_range(X, fr, cents, offs, assignments, start, end):   # KMeans.scala:24-55
    rng = JRandom(0)
    for i in range(start, end):
        row = X[i]
        mn = _FLT_MAX
        for k, c in enumerate(cents):
            d = f32(0)
            for j in range(len(c)):
                d = f32(d + f32(f32(row[j + fr]) * c[j]))
            d = f32(offs[k] - f32(f32(2) * d))
            if d < mn or (d == mn and rng.next_boolean()):
                assignments[i] = k
                mn = d


def kmeans_assign(X, fr, cents, rng_batch=0, assignments=None):
    offs = kmeans_offsets(cents)
    n = len(X)
    if assignments is None:
        assignments = [0] * n
    if rng_batch <= 0:
        kmeans_assign_range(X, fr, cents, offs, assignments, 0, n)   # KMeans.scala:70-98
    else:
        for b in range(0, n, rng_batch):                             # KMeans.scala:57-68
            kmeans_assign_range(X, fr, cents, offs, assignments, b, min(n, b + rng_batch))
    return assignments


def kmeans_from_assignment(X, fr, s, k, assignments):   # KMeans.scala:198-226
    cents = [[f32(0)] * s for _ in range(k)]
    counts = [0] * k
    for i, v in enumerate(X):
        a = assignments[i]
        c = cents[a]
        n = counts[a] + 1
        for j in range(s):
            p = c[j]
            c[j] = f32(p + f32(f32(f32(v[j + fr]) - p) / f32(n)))
        counts[a] = n
    return cents


def kmeans_compute_clusters(X, fr, un, k, max_iterations, seed=0):   # KMeans.scala:134-157
    s = un - fr
    prev = kmeans_init(X, fr, un, k, seed)
    pa = kmeans_assign(X, fr, prev, 25000)
    reports = [(0, False)]
    i = 0
    while i <= max_iterations:
        nxt = kmeans_from_assignment(X, fr, s, k, pa)
        na = kmeans_assign(X, fr, nxt, 25000)
        conv = pa == na
        reports.append((i, conv))
        i = max_iterations + 1 if conv else i + 1
        prev, pa = nxt, na
    return prev, reports


def prepare_query(quantizers, queries):                 # Index.scala:352-383
    # quantizers: list of (from, centroids)
    out = [[[f32(0)] * len(q[1]) for q in quantizers] for _ in queries]
    for j, (off, cents) in enumerate(quantizers):
        for i, c in enumerate(cents):
            for qi, query in enumerate(queries):
                s = f32(0)
                for t in range(len(c)):
                    d = f32(f32(query[t + off]) - c[t])
                    s = f32(s + f32(d * d))
                out[qi][j][i] = s
    return out


def pq_batch_query(quantizers, codes, n, queries, K, frm, until):   # Index.scala:417-440
    assert frm <= until and frm >= 0 and until <= n
    T = prepare_query(quantizers, queries)
    heaps = [Heap(K) for _ in queries]
    i = frm
    while i < until:
        bs = min(4096, until - i)
        for q, heap in enumerate(heaps):
            ds = [f32(0)] * bs                           # Index.scala:393-409
            for j in range(len(codes)):
                qds, code = T[q][j], codes[j]
                for r in range(bs):
                    ds[r] = f32(ds[r] + qds[code[i + r]])
            for r in range(bs):
                heap.update(i + r, ds[r])
        i += bs
    return [h.drain() for h in heaps]


def exact_knn(X, query, K, frm=0, until=None):          # Index.scala:209-229
    until = len(X) if until is None else until
    h = Heap(K)
    for i in range(frm, until):
        h.update(i, distance_sq(X[i], query))
    return h.drain()
