"""ctypes binding for the C oracle (oracle/gulon_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of gulon_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
PARITY UNPINNED vs the JVM reference (no JVM here, no golden vectors in the
reference); pinned by JDK Random KATs, CoderSpec KATs, ported properties and
the independent numpy restatement in py_oracle.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "libgulon_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "gulon_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


class _Report(C.Structure):
    _fields_ = [("num_iterations", C.c_int32), ("converged", C.c_int32),
                ("step_count", C.c_int32), ("step_mean", C.c_float), ("step_s", C.c_float)]


class _JRandom(C.Structure):
    _fields_ = [("seed", C.c_uint64)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.go_jr_init.argtypes = [C.POINTER(_JRandom), C.c_int64]
        L.go_jr_next_int.argtypes = [C.POINTER(_JRandom)]
        L.go_jr_next_int.restype = C.c_int32
        L.go_jr_next_int_bound.argtypes = [C.POINTER(_JRandom), C.c_int32]
        L.go_jr_next_int_bound.restype = C.c_int32
        L.go_jr_next_boolean.argtypes = [C.POINTER(_JRandom)]
        L.go_jr_next_boolean.restype = C.c_int32
        L.go_jr_next_float.argtypes = [C.POINTER(_JRandom)]
        L.go_jr_next_float.restype = C.c_float
        L.go_subvectors.argtypes = [C.c_int32, C.c_int32, _i32p, _i32p]
        L.go_distance_sq.argtypes = [_f32p, _f32p, C.c_int32]
        L.go_distance_sq.restype = C.c_float
        L.go_normalize.argtypes = [_f32p, C.c_int32, _f32p]
        L.go_grouped_query.argtypes = [_i32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f32p, _f32p, _i32p, C.c_int32,
                                       _f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i32p, _f32p, _i32p]
        L.go_grouped_query.restype = C.c_int32
        L.go_heap_new.argtypes = [C.c_int32]
        L.go_heap_new.restype = C.c_void_p
        L.go_heap_free.argtypes = [C.c_void_p]
        L.go_heap_size.argtypes = [C.c_void_p]
        L.go_heap_size.restype = C.c_int32
        L.go_heap_keys.argtypes = [C.c_void_p]
        L.go_heap_keys.restype = C.POINTER(C.c_int32)
        L.go_heap_values.argtypes = [C.c_void_p]
        L.go_heap_values.restype = C.POINTER(C.c_float)
        L.go_heap_delete.argtypes = [C.c_void_p]
        L.go_heap_delete.restype = C.c_int32
        L.go_heap_update.argtypes = [C.c_void_p, C.c_int32, C.c_float]
        L.go_heap_merge.argtypes = [C.c_void_p, C.c_void_p]
        L.go_heap_drain.argtypes = [C.c_void_p, _i32p, _f32p]
        L.go_heap_drain.restype = C.c_int32
        L.go_kmeans_offsets.argtypes = [_f32p, C.c_int32, C.c_int32, _f32p]
        L.go_kmeans_init.argtypes = [_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, _f32p, _i32p]
        L.go_kmeans_assign.argtypes = [_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f32p,
                                       C.c_int32, C.c_int32, _i32p]
        L.go_kmeans_from_assignment.argtypes = [_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                C.c_int32, _i32p, _f32p]
        L.go_kmeans_iterate.argtypes = [_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f32p,
                                        C.c_int32, C.c_int32, _f32p]
        L.go_kmeans_compute_clusters.argtypes = [_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                 C.c_int32, C.c_int32, C.c_int32, _f32p,
                                                 C.POINTER(_Report), C.c_int32]
        L.go_kmeans_compute_clusters.restype = C.c_int32
        L.go_coder_width_for_clusters.argtypes = [C.c_int32]
        L.go_coder_width_for_clusters.restype = C.c_int32
        L.go_coder_round_width.argtypes = [C.c_int32]
        L.go_coder_round_width.restype = C.c_int32
        L.go_coder_bytes.argtypes = [C.c_int32, C.c_int32]
        L.go_coder_bytes.restype = C.c_int32
        L.go_coder_build.argtypes = [C.c_int32, _i32p, C.c_int32, _u8p]
        L.go_coder_build.restype = C.c_int32
        L.go_coder_get.argtypes = [C.c_int32, _u8p, C.c_int32, C.c_int32]
        L.go_coder_get.restype = C.c_int32
        L.go_pq_train.argtypes = [_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                  _f32p, _i32p, _i32p]
        L.go_pq_encode.argtypes = [_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f32p, _i32p]
        L.go_pq_decode.argtypes = [_i32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f32p, _f32p]
        L.go_prepare_query.argtypes = [_f32p, C.c_int32, C.c_int32, C.c_int32, _f32p, C.c_int32, _f32p]
        L.go_pq_batch_query.argtypes = [_i32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f32p,
                                        _f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        _i32p, _f32p, _i32p]
        L.go_pq_batch_query.restype = C.c_int32
        L.go_pq_batch_query_u8.argtypes = [_u8p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f32p,
                                           _f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           _i32p, _f32p, _i32p]
        L.go_pq_batch_query_u8.restype = C.c_int32
        L.go_exact_knn.argtypes = [_f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f32p,
                                   C.c_int32, C.c_int32, _i32p, _f32p, _i32p]
        L.go_exact_knn.restype = C.c_int32
        L.go_recall.argtypes = [_f32p, C.c_int32, _f32p, C.c_int32, C.c_int32, _i32p, _i32p,
                                _f32p, _i32p, C.POINTER(C.c_float)]
        L.go_recall.restype = C.c_float
        L.go_synth_fill.argtypes = [_f32p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_uint64,
                                    C.c_int32]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class JavaRandom:
    """java.util.Random (48-bit LCG)."""

    def __init__(self, seed):
        self._r = _JRandom()
        lib().go_jr_init(C.byref(self._r), int(seed))

    def next_int(self, bound=None):
        if bound is None:
            return lib().go_jr_next_int(C.byref(self._r))
        return lib().go_jr_next_int_bound(C.byref(self._r), int(bound))

    def next_boolean(self):
        return bool(lib().go_jr_next_boolean(C.byref(self._r)))

    def next_float(self):
        return float(lib().go_jr_next_float(C.byref(self._r)))


def subvectors(d, m):
    fr = np.zeros(m, np.int32)
    un = np.zeros(m, np.int32)
    lib().go_subvectors(d, m, fr, un)
    return fr, un


def distance_sq(x, y):
    x, y = _f32(x), _f32(y)
    return np.float32(lib().go_distance_sq(x, y, x.size))


def normalize(x):
    x = _f32(x)
    out = np.empty_like(x)
    lib().go_normalize(x, x.size, out)
    return out


class TopKHeap:
    """TopKHeap.scala restated (keys Int, values Float, bounded max-heap)."""

    def __init__(self, k):
        self.k = k
        self._h = lib().go_heap_new(k)

    def __del__(self):
        try:
            lib().go_heap_free(self._h)
        except Exception:
            pass

    @property
    def size(self):
        return lib().go_heap_size(self._h)

    def update(self, key, value):
        lib().go_heap_update(self._h, int(key), float(np.float32(value)))

    def delete(self):
        r = lib().go_heap_delete(self._h)
        if r == -(2 ** 31):
            raise RuntimeError("heap is empty")
        return r

    def merge(self, other):
        lib().go_heap_merge(self._h, other._h)

    def raw(self):
        n = self.size
        ks = np.ctypeslib.as_array(lib().go_heap_keys(self._h), (max(self.k, 1),))[:n].copy()
        vs = np.ctypeslib.as_array(lib().go_heap_values(self._h), (max(self.k, 1),))[:n].copy()
        return ks, vs

    def drain(self):
        ks = np.zeros(max(self.k, 1), np.int32)
        vs = np.zeros(max(self.k, 1), np.float32)
        n = lib().go_heap_drain(self._h, ks, vs)
        return ks[:n], vs[:n]


def kmeans_offsets(Cn):
    Cn = _f32(Cn)
    off = np.zeros(Cn.shape[0], np.float32)
    lib().go_kmeans_offsets(Cn, Cn.shape[0], Cn.shape[1], off)
    return off


def kmeans_init(X, frm, s, k, seed=0):
    X = _f32(X)
    n, ld = X.shape
    Cn = np.zeros((k, s), np.float32)
    rows = np.zeros(k, np.int32)
    lib().go_kmeans_init(X, n, ld, frm, s, k, seed, Cn, rows)
    return Cn, rows


def kmeans_assign(X, frm, s, Cn, rng_batch=0, assignments=None):
    """rng_batch=0: serial assign (KMeans.scala:70-98); 25000: parAssign (:57-68)."""
    X, Cn = _f32(X), _f32(Cn)
    n, ld = X.shape
    if assignments is None:
        assignments = np.zeros(n, np.int32)
    lib().go_kmeans_assign(X, n, ld, frm, s, Cn, Cn.shape[0], rng_batch, assignments)
    return assignments


def kmeans_from_assignment(X, frm, s, k, assignments):
    X = _f32(X)
    n, ld = X.shape
    Cn = np.zeros((k, s), np.float32)
    lib().go_kmeans_from_assignment(X, n, ld, frm, s, k, _i32(assignments), Cn)
    return Cn


def kmeans_iterate(X, frm, s, Cn, iters):
    X, Cn = _f32(X), _f32(Cn)
    n, ld = X.shape
    out = np.zeros_like(Cn)
    lib().go_kmeans_iterate(X, n, ld, frm, s, Cn, Cn.shape[0], iters, out)
    return out


def kmeans_compute_clusters(X, frm, s, k, max_iterations, seed=0):
    X = _f32(X)
    n, ld = X.shape
    Cn = np.zeros((k, s), np.float32)
    maxrep = max_iterations + 3
    reps = (_Report * maxrep)()
    nrep = lib().go_kmeans_compute_clusters(X, n, ld, frm, s, k, max_iterations, seed, Cn, reps, maxrep)
    out = [dict(num_iterations=r.num_iterations, converged=bool(r.converged), step_count=r.step_count,
                step_mean=np.float32(r.step_mean), step_s=np.float32(r.step_s)) for r in reps[:nrep]]
    return Cn, out


def coder_width_for_clusters(k):
    return lib().go_coder_width_for_clusters(k)


def coder_round_width(w):
    return lib().go_coder_round_width(w)


def coder_bytes(width, n):
    return lib().go_coder_bytes(width, n)


def coder_build(width, idx):
    idx = _i32(idx)
    nb = coder_bytes(width, idx.size)
    code = np.zeros(max(nb, 1), np.uint8)
    lib().go_coder_build(width, idx, idx.size, code)
    return code[:nb]


def coder_get(width, code, n, i):
    c = np.ascontiguousarray(code, np.uint8)
    if c.size == 0:
        c = np.zeros(1, np.uint8)
    return lib().go_coder_get(width, c, n, i)


def pq_train(X, m, k, max_iterations):
    X = _f32(X)
    n, d = X.shape
    cents = np.zeros(k * d, np.float32)
    iters = np.zeros(m, np.int32)
    conv = np.zeros(m, np.int32)
    lib().go_pq_train(X, n, d, m, k, max_iterations, cents, iters, conv)
    return cents, iters, conv


def pq_encode(X, m, k, cents):
    X = _f32(X)
    n, d = X.shape
    idx = np.zeros((m, n), np.int32)
    lib().go_pq_encode(X, n, d, m, k, _f32(cents), idx)
    return idx


def pq_decode(idx, d, k, cents):
    idx = _i32(idx)
    m, n = idx.shape
    X = np.zeros((n, d), np.float32)
    lib().go_pq_decode(idx, n, d, m, k, _f32(cents), X)
    return X


def prepare_query(cents, d, m, k, Q):
    Q = _f32(Q)
    B = Q.shape[0]
    T = np.zeros((B, m, k), np.float32)
    lib().go_prepare_query(_f32(cents), d, m, k, Q, B, T)
    return T


def pq_batch_query(idx, d, k, cents, Q, K, from_row=0, until_row=None):
    Q = _f32(Q)
    B = Q.shape[0]
    if np.asarray(idx).dtype == np.uint8:
        codes = np.ascontiguousarray(idx)
        m, n = codes.shape
        fn = lib().go_pq_batch_query_u8
        arr = codes
    else:
        arr = _i32(idx)
        m, n = arr.shape
        fn = lib().go_pq_batch_query
    if until_row is None:
        until_row = n
    oi = np.zeros((B, max(K, 1)), np.int32)
    od = np.zeros((B, max(K, 1)), np.float32)
    oc = np.zeros(B, np.int32)
    rc = fn(arr, n, d, m, k, _f32(cents), Q, B, K, from_row, until_row, oi, od, oc)
    if rc != 0:
        raise ValueError("requirement failed")
    return oi, od, oc


def exact_knn(X, Q, K, from_row=0, until_row=None):
    X, Q = _f32(X), _f32(Q)
    n, d = X.shape
    B = Q.shape[0]
    if until_row is None:
        until_row = n
    oi = np.zeros((B, max(K, 1)), np.int32)
    od = np.zeros((B, max(K, 1)), np.float32)
    oc = np.zeros(B, np.int32)
    rc = lib().go_exact_knn(X, n, d, from_row, until_row, Q, B, K, oi, od, oc)
    if rc != 0:
        raise ValueError("requirement failed")
    return oi, od, oc


def group_rows(assignments, coarse_centroids, word_order=None):
    """WordVectors.grouped (WordVectors.scala:24-58), restated literally on row ids.

    `assignments` = clustering.parAssign over the rows in their ORIGINAL order (:27); `word_order` = the
    row indices stably sorted by word (`Array.range(0, size).sortBy(word(_))`, :28-29; None: the rows are
    already in word order); then the stable `.sortBy(assignments(_))` (:30) and the builder loop (:37-52),
    INCLUDING its seed `prev = assignments(0)` (:38-39): that is the cluster of ORIGINAL row 0, not of the
    first grouped row, so whenever row 0 is not in the lowest-numbered non-empty cluster the reference
    emits a leading EMPTY group [0, 0) -- offsets(0) == 0 -- carrying a copy of row 0's centroid.
    Returns (perm, group_centroids [g][d], offsets [g-1]) with g == len(offsets) + 1."""
    a = np.asarray(assignments, np.int32)
    n = len(a)
    C_ = _f32(coarse_centroids)
    order = np.arange(n) if word_order is None else np.asarray(word_order)
    indices = order[np.argsort(a[order], kind="stable")]        # :28-30
    offsets, cents = [], []
    if n > 0:                                                   # :36
        prev = int(a[0])                                        # :38
        cents.append(C_[prev])                                  # :39
        for i in range(n):                                      # :40-51
            j = int(indices[i])
            aj = int(a[j])
            if prev != aj:
                offsets.append(i)
                prev = aj
                cents.append(C_[prev])
    gc = np.ascontiguousarray(np.stack(cents)) if cents else C_[:0]
    return indices.astype(np.int32), gc, np.asarray(offsets, np.int32)


def group_residuals(X, perm, group_centroids, offsets):
    """WordVectors.Grouped.residuals (WordVectors.scala:118-138): grouped row - its group's centroid."""
    X = _f32(X)[perm]
    n = X.shape[0]
    bounds = np.r_[0, np.asarray(offsets, np.int64), n]
    out = np.empty_like(X)
    for c in range(len(bounds) - 1):
        out[bounds[c]:bounds[c + 1]] = X[bounds[c]:bounds[c + 1]] - _f32(group_centroids)[c]
    return out


def grouped_query(idx, d, k, pq_cents, group_centroids, offsets, Q, K, strategy, limit):
    """GroupedIndex.batchQuery (Index.scala:254-282); strategy 0 = LimitGroups, 1 = LimitVectors."""
    Q = _f32(Q)
    B = Q.shape[0]
    arr = _i32(idx)
    m, n = arr.shape
    gc = _f32(group_centroids)
    g = gc.shape[0]
    off = _i32(offsets) if len(offsets) else np.zeros(1, np.int32)
    oi = np.zeros((B, max(K, 1)), np.int32)
    od = np.zeros((B, max(K, 1)), np.float32)
    oc = np.zeros(B, np.int32)
    rc = lib().go_grouped_query(arr, n, d, m, k, _f32(pq_cents), gc, off, g, Q, B, K, strategy, limit, oi, od, oc)
    if rc != 0:
        raise ValueError("requirement failed")
    return oi, od, oc


def recall(X, Q, K, ann_idx, ann_count, exact_dist, exact_count):
    X, Q = _f32(X), _f32(Q)
    sd = C.c_float(0)
    mean = lib().go_recall(X, X.shape[1], Q, Q.shape[0], K, _i32(ann_idx), _i32(ann_count),
                           _f32(exact_dist), _i32(exact_count), C.byref(sd))
    return float(mean), float(sd.value)


def synth(n, d, kind, seed, ncentres=1000, row0=0):
    """kind: 0 iid N(0,1), 1 clustered, 2 U[0,1), 3 overlapping clusters.  Bit-identical to the HIP generator."""
    X = np.zeros((n, d), np.float32)
    lib().go_synth_fill(X, row0, n, d, kind, seed, ncentres)
    return X
