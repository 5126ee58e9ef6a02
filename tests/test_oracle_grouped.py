"""GroupedIndex in the CPU oracle: the reference's own property (WordVectorsSpec.scala:109-123), the
grouping invariants of WordVectors.grouped (WordVectors.scala:24-58), and an independent numpy
restatement of GroupedIndex.query (Index.scala:265-299) that must agree with the C one bit for bit
on tie-free data."""
import numpy as np
import pytest

from conftest import bits


def _data(seed, n, d, groups):
    rng = np.random.default_rng(seed)
    X = (rng.standard_normal((n, d)) + 4.0 * rng.integers(0, 3, (n, 1))).astype(np.float32)
    return X


def _pipeline(oracle, X, groups, m, k, iters=3):
    n, d = X.shape
    C, _ = oracle.kmeans_compute_clusters(X, 0, d, groups, iters)
    assign = oracle.kmeans_assign(X, 0, d, C, rng_batch=25000)
    perm, cents, offsets = oracle.group_rows(assign, C)
    R = oracle.group_residuals(X, perm, cents, offsets)
    pq_cents = oracle.pq_train(R, m, k, iters)[0]
    codes = oracle.pq_encode(R, m, k, pq_cents)
    return assign, perm, cents, offsets, R, pq_cents, codes


@pytest.mark.parametrize("seed,n,d,groups", [(1, 500, 6, 7), (2, 1200, 10, 25), (3, 64, 4, 64)])
def test_grouping_invariants_and_residuals(oracle, seed, n, d, groups):
    X = _data(seed, n, d, groups)
    C, _ = oracle.kmeans_compute_clusters(X, 0, d, groups, 3)
    assign = oracle.kmeans_assign(X, 0, d, C, rng_batch=25000)
    perm, cents, offsets = oracle.group_rows(assign, C)
    assert sorted(perm.tolist()) == list(range(n))
    ga = assign[perm]
    assert np.all(np.diff(ga) >= 0)                                   # grouped by cluster, ascending
    for c in np.unique(ga):                                           # stable: original order inside a group
        assert np.all(np.diff(perm[ga == c]) > 0)
    bounds = np.r_[0, offsets, n]
    assert len(cents) == len(offsets) + 1 == len(np.unique(assign))   # only non-empty clusters
    for gi in range(len(cents)):
        rows = perm[bounds[gi]:bounds[gi + 1]]
        assert len(rows) > 0 and len(set(assign[rows])) == 1
        assert np.array_equal(bits(cents[gi]), bits(C[assign[rows[0]]]))
    R = oracle.group_residuals(X, perm, cents, offsets)
    # WordVectorsSpec.scala:109-123: centroid + residual is (nearly) the vector
    back = np.concatenate([R[bounds[gi]:bounds[gi + 1]] + cents[gi] for gi in range(len(cents))])
    assert np.allclose(back, X[perm], atol=0.05)


def _numpy_grouped_query(oracle, codes, d, k, pq_cents, cents, offsets, q, K, strategy, limit):
    """Independent restatement with numpy sorts (valid where no two distances tie)."""
    m, n = codes.shape
    g = len(cents)
    bounds = np.r_[0, offsets, n]
    cd = np.array([oracle.distance_sq(cents[c], q) for c in range(g)], np.float32)
    order = np.argsort(cd, kind="stable")
    if strategy == 0:
        order = order[:limit]
    else:
        cnt, i = 0, 0
        while i < g and cnt < limit:
            cnt += bounds[order[i] + 1] - bounds[order[i]]
            i += 1
        order = order[:i]
    cand = []
    for c in order:
        res = (q - cents[c]).astype(np.float32)
        T = oracle.prepare_query(pq_cents, d, m, k, res[None, :])[0]
        acc = np.zeros(bounds[c + 1] - bounds[c], np.float32)
        for j in range(m):
            acc = (acc + T[j, codes[j, bounds[c]:bounds[c + 1]]]).astype(np.float32)
        cand += [(float(v), int(bounds[c] + r)) for r, v in enumerate(acc)]
    cand.sort()
    return cand[:K]


@pytest.mark.parametrize("strategy,limit", [(0, 2), (0, 5), (1, 150), (1, 10 ** 6)])
def test_grouped_query_matches_numpy_restatement(oracle, strategy, limit):
    X = _data(11, 900, 8, 9)
    assign, perm, cents, offsets, R, pq_cents, codes = _pipeline(oracle, X, 9, 4, 16)
    rng = np.random.default_rng(5)
    Q = (X[rng.integers(0, len(X), 6)] + 0.01 * rng.standard_normal((6, 8))).astype(np.float32)
    K = 7
    oi, od, oc = oracle.grouped_query(codes, 8, 16, pq_cents, cents, offsets, Q, K, strategy, limit)
    for qi in range(len(Q)):
        exp = _numpy_grouped_query(oracle, codes, 8, 16, pq_cents, cents, offsets, Q[qi], K, strategy, limit)
        vals = [v for v, _ in exp]
        assert oc[qi] == len(exp)
        assert od[qi, :oc[qi]].tolist() == vals                        # distances always
        if len(set(vals)) == len(vals):                                # ids wherever nothing ties
            assert oi[qi, :oc[qi]].tolist() == [r for _, r in exp]


def test_limit_groups_all_equals_scanning_every_group(oracle):
    """LimitGroups(g) and LimitVectors(n) search everything: same distances as each other."""
    X = _data(21, 700, 6, 8)
    assign, perm, cents, offsets, R, pq_cents, codes = _pipeline(oracle, X, 8, 3, 8)
    Q = X[:5]
    a = oracle.grouped_query(codes, 6, 8, pq_cents, cents, offsets, Q, 10, 0, len(cents))
    b = oracle.grouped_query(codes, 6, 8, pq_cents, cents, offsets, Q, 10, 1, len(X))
    assert np.array_equal(bits(a[1]), bits(b[1])) and np.array_equal(a[2], b[2])
