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
    # only non-empty clusters -- plus the reference's leading empty group when ORIGINAL row 0 is not in the
    # lowest-numbered non-empty cluster (`prev = assignments(0)`, WordVectors.scala:38-39)
    lead = int(assign[0] != assign.min())
    assert len(cents) == len(offsets) + 1 == len(np.unique(assign)) + lead
    if lead:
        assert offsets[0] == 0 and np.array_equal(bits(cents[0]), bits(C[assign[0]]))
    for gi in range(lead, len(cents)):
        rows = perm[bounds[gi]:bounds[gi + 1]]
        assert len(rows) > 0 and len(set(assign[rows])) == 1
        assert np.array_equal(bits(cents[gi]), bits(C[assign[rows[0]]]))
    R = oracle.group_residuals(X, perm, cents, offsets)
    # WordVectorsSpec.scala:109-123: centroid + residual is (nearly) the vector
    back = np.concatenate([R[bounds[gi]:bounds[gi + 1]] + cents[gi] for gi in range(len(cents))])
    assert np.allclose(back, X[perm], atol=0.05)


def test_group_rows_is_the_reference_loop(oracle):
    """Hand-worked cases of WordVectors.scala:24-58."""
    C = np.arange(8, dtype=np.float32).reshape(4, 2)
    # row 0 in the lowest non-empty cluster: no empty group
    perm, cents, off = oracle.group_rows(np.array([1, 3, 1, 3, 2], np.int32), C)
    assert perm.tolist() == [0, 2, 4, 1, 3] and off.tolist() == [2, 3] and cents.tolist() == C[[1, 2, 3]].tolist()
    # row 0 in cluster 3, lowest non-empty cluster is 1: leading empty group with centroid 3, which comes again later
    perm, cents, off = oracle.group_rows(np.array([3, 1, 2, 1, 3], np.int32), C)
    assert perm.tolist() == [1, 3, 2, 0, 4] and off.tolist() == [0, 2, 3] and cents.tolist() == C[[3, 1, 2, 3]].tolist()
    # word order: rows sorted by word first (stable), then by cluster; prev still comes from ORIGINAL row 0
    perm, cents, off = oracle.group_rows(np.array([2, 0, 2, 0], np.int32), C, word_order=[3, 2, 1, 0])
    assert perm.tolist() == [3, 1, 2, 0] and off.tolist() == [0, 2] and cents.tolist() == C[[2, 0, 2]].tolist()
    # one cluster, no rows
    perm, cents, off = oracle.group_rows(np.array([2, 2, 2], np.int32), C)
    assert perm.tolist() == [0, 1, 2] and off.tolist() == [] and cents.tolist() == C[[2]].tolist()
    perm, cents, off = oracle.group_rows(np.zeros(0, np.int32), C)
    assert len(perm) == 0 and len(off) == 0 and len(cents) == 0


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
