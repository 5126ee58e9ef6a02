"""The C oracle and the independent numpy.float32 Python restatement must agree
bit for bit on small inputs (both are written from the Scala source)."""
import numpy as np
import pytest

from conftest import bits
from oracle import py_oracle as po


def _data(seed, n, d, scale=1.0, dup=False):
    rng = np.random.default_rng(seed)
    X = (rng.standard_normal((n, d)) * scale).astype(np.float32)
    if dup:                      # duplicate rows => duplicate init centroids => tie-break RNG
        X[n // 2:] = X[: n - n // 2]
    return X


@pytest.mark.parametrize("seed,n,d,fr,un,k,dup", [
    (0, 200, 6, 0, 6, 5, False), (1, 300, 9, 3, 7, 8, False), (2, 120, 4, 1, 3, 16, True),
    (3, 64, 3, 0, 1, 4, True)])
def test_kmeans_pieces(oracle, seed, n, d, fr, un, k, dup):
    X = _data(seed, n, d, dup=dup)
    s = un - fr
    C0, rows = oracle.kmeans_init(X, fr, s, k, seed)
    P0 = po.kmeans_init(X.tolist(), fr, un, k, seed)
    assert np.array_equal(bits(C0), bits(np.array(P0, np.float32)))
    assert np.array_equal(bits(oracle.kmeans_offsets(C0)), bits(np.array(po.kmeans_offsets(P0), np.float32)))
    for rb in (0, 50):
        a = oracle.kmeans_assign(X, fr, s, C0, rb)
        b = po.kmeans_assign(X.tolist(), fr, P0, rb)
        assert a.tolist() == b
    C1 = oracle.kmeans_from_assignment(X, fr, s, k, a)
    P1 = po.kmeans_from_assignment(X.tolist(), fr, s, k, b)
    assert np.array_equal(bits(C1), bits(np.array(P1, np.float32)))


def test_compute_clusters(oracle):
    X = _data(5, 400, 8, dup=True)
    for (fr, un, k, it, seed) in [(0, 4, 6, 5, 0), (4, 8, 3, 20, 1)]:
        Cc, reps = oracle.kmeans_compute_clusters(X, fr, un - fr, k, it, seed)
        Pc, preps = po.kmeans_compute_clusters(X.tolist(), fr, un, k, it, seed)
        assert np.array_equal(bits(Cc), bits(np.array(Pc, np.float32)))
        assert [(r["num_iterations"], r["converged"]) for r in reps] == preps


def test_tie_break_stream_crosses_batches(oracle):
    # all-zero centroids but one: every row ties k-1 times; RNG restarts per batch
    rng = np.random.default_rng(7)
    X = rng.standard_normal((130, 2)).astype(np.float32)
    C = np.zeros((4, 2), np.float32)
    for rb in (0, 25, 64):
        a = oracle.kmeans_assign(X, 0, 2, C, rb)
        b = po.kmeans_assign(X.tolist(), 0, C.tolist(), rb)
        assert a.tolist() == b
    assert len(set(oracle.kmeans_assign(X, 0, 2, C, 0).tolist())) > 1


@pytest.mark.parametrize("n,d,m,k,B,K,fr,un", [(300, 10, 4, 7, 3, 5, 0, 300), (5000, 6, 3, 16, 2, 10, 100, 4700),
                                                (20, 4, 2, 3, 2, 30, 0, 20), (50, 5, 5, 1, 1, 4, 10, 10)])
def test_query_path(oracle, n, d, m, k, B, K, fr, un):
    rng = np.random.default_rng(n)
    cents = rng.standard_normal(k * d).astype(np.float32)
    idx = rng.integers(0, k, (m, n)).astype(np.int32)
    Q = rng.standard_normal((B, d)).astype(np.float32)
    T = oracle.prepare_query(cents, d, m, k, Q)
    sub = po.subvectors(d, m)
    quant = [(f, [[np.float32(v) for v in cents[k * f + c * (u - f): k * f + (c + 1) * (u - f)]]
                  for c in range(k)]) for f, u in sub]
    PT = po.prepare_query(quant, Q.tolist())
    assert np.array_equal(bits(T), bits(np.array(PT, np.float32)))
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K, fr, un)
    pres = po.pq_batch_query(quant, idx.tolist(), n, Q.tolist(), K, fr, un)
    for q in range(B):
        ks, vs = pres[q]
        assert oc[q] == len(ks) == min(K, un - fr)
        assert oi[q, :oc[q]].tolist() == ks
        assert np.array_equal(bits(od[q, :oc[q]]), bits(np.array(vs, np.float32)))


def test_exact_knn(oracle):
    X = _data(11, 500, 7)
    Q = _data(12, 3, 7)
    oi, od, oc = oracle.exact_knn(X, Q, 10, 20, 480)
    for q in range(3):
        ks, vs = po.exact_knn(X.tolist(), Q[q].tolist(), 10, 20, 480)
        assert oi[q].tolist() == ks
        assert np.array_equal(bits(od[q]), bits(np.array(vs, np.float32)))
