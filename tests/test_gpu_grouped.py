"""GPU parity tests for GroupedIndex (Index.scala:231-308): coarse groups + PQ on residuals.
Everything through the C ABI; the CPU oracle restates GroupedIndex.query literally (per-group
TopKHeap, TopKHeap.merge in array order), and the GPU kernels run the same literal heaps, so ids,
order and distances must be equal bit for bit -- also on data with exact distance ties."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gulon_amd
    assert gulon_amd.native.device_count() >= 1
    return gulon_amd


def _build(oracle, g, n, d, groups, m, k, seed, dup=0, iters=3):
    rng = np.random.default_rng(seed)
    X = (rng.standard_normal((n, d)) + 3.0 * rng.integers(0, 4, (n, 1))).astype(np.float32)
    if dup:
        X[-dup:] = X[:dup]                                   # exact duplicates: distance ties everywhere
    dm = g.DeviceMatrix.from_host(X)
    coarse = g.KMeans.compute_clusters(g.Vectors(dm), g.KMeansConfig(groups, iters))
    gv = g.group(dm, coarse)
    pq = g.ProductQuantizer.apply(gv.residuals, g.ProductQuantizerConfig(k, m, iters))
    return X, dm, coarse, gv, pq


def _oracle_side(oracle, X, coarse, gv, pq, n):
    """The same pipeline on the CPU: grouping, residuals, codes -- checked against the GPU's."""
    assign = oracle.kmeans_assign(X, 0, X.shape[1], coarse.centroids, rng_batch=25000)
    perm, cents, offsets = oracle.group_rows(assign, coarse.centroids)
    assert np.array_equal(perm, gv.perm) and np.array_equal(offsets, gv.offsets)
    assert np.array_equal(bits(cents), bits(gv.centroids))
    R = oracle.group_residuals(X, perm, cents, offsets)
    assert np.array_equal(bits(R), bits(gv.residuals.get_rows(np.arange(n, dtype=np.int32))))
    return R, cents, offsets


@pytest.mark.parametrize("strategy,limit", [("groups", 1), ("groups", 3), ("groups", 8), ("vectors", 500),
                                            ("vectors", 1), ("vectors", 10 ** 9), ("groups", 100)])
@pytest.mark.parametrize("n,d,groups,m,k,B,K,dup", [
    (6000, 16, 12, 4, 16, 9, 5, 0),
    (20000, 32, 40, 8, 256, 17, 10, 0),
    (8000, 24, 10, 6, 64, 5, 10, 1500),          # duplicated rows: ties inside and across groups
])
@pytest.mark.parametrize("qm", [False, True])
def test_grouped_query_equals_reference(oracle, g, monkeypatch, n, d, groups, m, k, B, K, dup, strategy, limit, qm):
    if qm:      # the quantizer-major kernel normally takes over from 16 searched groups per query on
        monkeypatch.setenv("GULON_GROUPED_QM", "1")
    X, dm, coarse, gv, pq = _build(oracle, g, n, d, groups, m, k, seed=n + d, dup=dup)
    R, cents, offsets = _oracle_side(oracle, X, coarse, gv, pq, n)
    strat = g.LimitGroups(limit) if strategy == "groups" else g.LimitVectors(limit)
    index = g.Index.grouped(gv, pq, strat)
    codes = index.data.indices()
    assert np.array_equal(codes, oracle.pq_encode(R, m, k, pq.flat_centroids()))
    rng = np.random.default_rng(1)
    Q = np.concatenate([X[rng.integers(0, n, B - 2)], (rng.standard_normal((2, d)) * 2).astype(np.float32)])
    oi, od, oc = index.batch_query_raw(K, Q)
    ei, ed, ec = oracle.grouped_query(codes, d, k, pq.flat_centroids(), cents, offsets, Q, K,
                                      0 if strategy == "groups" else 1, limit)
    assert np.array_equal(oc, ec)
    for q in range(B):
        assert oi[q, :oc[q]].tolist() == ei[q, :ec[q]].tolist()
        assert np.array_equal(bits(od[q, :oc[q]]), bits(ed[q, :ec[q]]))
    index.close()


@pytest.mark.parametrize("n,d,groups,m,k,limit", [
    (20000, 16, 3, 4, 256, 3),        # ~6700-row groups: several passes of 32 row blocks
    (9000, 21, 30, 7, 40, 12),        # odd m: the last step has one quantizer; 3-wide sub-vectors
    (6000, 50, 20, 5, 256, 20),       # 10-wide sub-vectors
    (6000, 64, 8, 4, 100, 8),         # 16-wide sub-vectors
    (30000, 32, 200, 8, 256, 120),    # the regime where it is picked without being forced
])
def test_quantizer_major_scan_equals_reference(oracle, g, monkeypatch, n, d, groups, m, k, limit):
    monkeypatch.setenv("GULON_GROUPED_QM", "1")
    B, K = 7, 10
    X, dm, coarse, gv, pq = _build(oracle, g, n, d, groups, m, k, seed=n + m, dup=600, iters=2)
    R, cents, offsets = _oracle_side(oracle, X, coarse, gv, pq, n)
    index = g.Index.grouped(gv, pq, g.LimitGroups(limit))
    rng = np.random.default_rng(3)
    Q = np.concatenate([X[rng.integers(0, n, B - 1)], (rng.standard_normal((1, d)) * 2).astype(np.float32)])
    oi, od, oc = index.batch_query_raw(K, Q)
    ei, ed, ec = oracle.grouped_query(index.data.indices(), d, k, pq.flat_centroids(), cents, offsets, Q, K, 0, limit)
    assert np.array_equal(oc, ec)
    for q in range(B):
        assert oi[q, :oc[q]].tolist() == ei[q, :ec[q]].tolist()
        assert np.array_equal(bits(od[q, :oc[q]]), bits(ed[q, :ec[q]]))
    index.close()


def test_single_group_and_few_rows(oracle, g):
    """One coarse cluster (offsets empty) and groups smaller than K."""
    n, d, m, k, K = 300, 8, 2, 8, 20
    X, dm, coarse, gv, pq = _build(oracle, g, n, d, 1, m, k, seed=5)
    assert len(gv.offsets) == 0 and len(gv.centroids) == 1
    index = g.Index.grouped(gv, pq, g.LimitGroups(4))
    Q = X[:7]
    oi, od, oc = index.batch_query_raw(K, Q)
    ei, ed, ec = oracle.grouped_query(index.data.indices(), d, k, pq.flat_centroids(), gv.centroids, gv.offsets, Q, K,
                                      0, 4)
    assert np.array_equal(oc, ec) and np.array_equal(oi, ei) and np.array_equal(bits(od), bits(ed))
    index.close()
    X, dm, coarse, gv, pq = _build(oracle, g, 40, d, 12, m, k, seed=6)      # ~3 rows per group
    index = g.Index.grouped(gv, pq, g.LimitGroups(3))
    oi, od, oc = index.batch_query_raw(K, X[:5])
    ei, ed, ec = oracle.grouped_query(index.data.indices(), d, k, pq.flat_centroids(), gv.centroids, gv.offsets,
                                      X[:5], K, 0, 3)
    assert np.array_equal(oc, ec)
    for q in range(5):
        assert oi[q, :oc[q]].tolist() == ei[q, :ec[q]].tolist()
    index.close()


def test_cosine_metric_normalises_the_query(oracle, g):
    n, d, m, k, K = 5000, 16, 4, 32, 10
    X, dm, coarse, gv, pq = _build(oracle, g, n, d, 8, m, k, seed=9)
    index = g.Index.grouped(gv, pq, g.LimitGroups(3), metric="cosine")
    Q = X[:6] * 7.5
    oi, od, oc = index.batch_query_raw(K, Q)
    Qn = np.stack([oracle.normalize(r) for r in Q])
    ei, ed, ec = oracle.grouped_query(index.data.indices(), d, k, pq.flat_centroids(), gv.centroids, gv.offsets, Qn,
                                      K, 0, 3)
    assert np.array_equal(oi, ei) and np.array_equal(bits(od), bits(ed))
    index.close()


def test_bad_arguments(g):
    import ctypes as C
    from gulon_amd import native as N
    h = C.c_void_p()
    codes = np.zeros(4 * 10, np.uint8)
    cents = np.zeros(256 * 8, np.float32)
    gc = np.zeros(2 * 8, np.float32)
    off = np.array([20], np.int32)                       # beyond n = 10
    assert N.lib().gulon_grouped_index_create(codes, 10, 8, 4, 256, cents, gc, off, 2, C.byref(h)) == -1


@pytest.mark.parametrize("n,d,groups,m,k,limit,dup,bad_query", [
    (120000, 32, 400, 16, 256, 120, 0, False),     # the regime it is built for: many groups, m = 16
    (60000, 24, 300, 8, 64, 90, 3000, False),      # duplicated rows: equal D~ at the cut, ties in the re-ranking
    (60000, 16, 250, 4, 16, 250, 0, True),         # every group searched; a NaN query and a far-away query
    (66000, 8, 11000, 4, 16, 100, 0, False),       # more groups than the group selection keeps keys in registers for
    (90000, 16, 600, 8, 64, 40, 2500, False),      # LimitGroups(<= 63) over many groups (the wavefront heap), duplicated rows
])
def test_by_group_filter_equals_reference(oracle, g, monkeypatch, capfd, n, d, groups, m, k, limit, dup, bad_query):
    """GroupedIndex.query (Index.scala:265-299) through the by-group pre-selection with 8-bit bound tables
    (grouped_filter.hip): it only chooses which rows are re-scored with the reference's arithmetic, so the answers must
    be the reference's; the statistics line shows that the path ran."""
    monkeypatch.setenv("GULON_GROUPED_STATS", "1")
    B, K = 40, 10
    X, dm, coarse, gv, pq = _build(oracle, g, n, d, groups, m, k, seed=n + groups, dup=dup, iters=2)
    R, cents, offsets = _oracle_side(oracle, X, coarse, gv, pq, n)
    index = g.Index.grouped(gv, pq, g.LimitGroups(limit))
    rng = np.random.default_rng(5)
    Q = np.concatenate([X[rng.integers(0, n, B - 2)], (rng.standard_normal((2, d)) * 3).astype(np.float32)])
    if bad_query:
        Q[3, 1] = np.nan
        Q[4] *= np.float32(1e4)
    oi, od, oc = index.batch_query_raw(K, Q)
    assert "by-group filter" in capfd.readouterr().err
    ei, ed, ec = oracle.grouped_query(index.data.indices(), d, k, pq.flat_centroids(), cents, offsets, Q, K, 0, limit)
    assert np.array_equal(oc, ec)
    for q in range(B):
        assert oi[q, :oc[q]].tolist() == ei[q, :ec[q]].tolist(), q
        assert np.array_equal(bits(od[q, :oc[q]]), bits(ed[q, :ec[q]])), q
    index.close()


def test_by_group_filter_with_hundreds_of_ties_at_the_cut(oracle, g, monkeypatch, capfd):
    """700 copies of one row: for a query at that row they all share the smallest D~, more entries at the cut than the
    survivors' pass orders (GF_PLACED = 512) -- the query must go to the literal kernels and come back with the
    reference's answer (TopKHeap order among equal distances, Index.scala:265-299)."""
    monkeypatch.setenv("GULON_GROUPED_STATS", "1")
    n, d, groups, m, k, limit, B, K = 40000, 16, 200, 8, 64, 60, 20, 10
    rng = np.random.default_rng(77)
    X = (rng.standard_normal((n, d)) + 3.0 * rng.integers(0, 4, (n, 1))).astype(np.float32)
    X[5000:5700] = X[17]
    dm = g.DeviceMatrix.from_host(X)
    coarse = g.KMeans.compute_clusters(g.Vectors(dm), g.KMeansConfig(groups, 2))
    gv = g.group(dm, coarse)
    pq = g.ProductQuantizer.apply(gv.residuals, g.ProductQuantizerConfig(k, m, 2))
    R, cents, offsets = _oracle_side(oracle, X, coarse, gv, pq, n)
    index = g.Index.grouped(gv, pq, g.LimitGroups(limit))
    Q = np.concatenate([X[17:18], X[5100:5101], X[rng.integers(0, n, B - 2)]])
    oi, od, oc = index.batch_query_raw(K, Q)
    assert "by-group filter" in capfd.readouterr().err
    ei, ed, ec = oracle.grouped_query(index.data.indices(), d, k, pq.flat_centroids(), cents, offsets, Q, K, 0, limit)
    assert np.array_equal(oc, ec)
    for q in range(B):
        assert oi[q, :oc[q]].tolist() == ei[q, :ec[q]].tolist(), q
        assert np.array_equal(bits(od[q, :oc[q]]), bits(ed[q, :ec[q]])), q
    index.close()


@pytest.mark.parametrize("n,d,groups,m,k,strategy,limit,K,dup", [
    (6000, 16, 12, 4, 16, "groups", 5, 64, 0),          # just past the wavefront heaps
    (20000, 32, 40, 16, 256, "groups", 12, 100, 0),
    (8000, 24, 10, 6, 64, "vectors", 3000, 200, 1500),  # duplicated rows: the heaps' array order decides
    (3000, 8, 6, 2, 8, "groups", 6, 1000, 0),           # Tests.scala's largest k; groups smaller than k
])
def test_more_than_63_neighbours(oracle, g, n, d, groups, m, k, strategy, limit, K, dup):
    """GroupedIndex.query with k > 63 (Tests.scala samples k up to 1000): the literal TopKHeaps with their arrays in LDS,
    merged in search order (Index.scala:265-299, TopKHeap.scala)."""
    B = 6
    X, dm, coarse, gv, pq = _build(oracle, g, n, d, groups, m, k, seed=n + K, dup=dup)
    R, cents, offsets = _oracle_side(oracle, X, coarse, gv, pq, n)
    strat = g.LimitGroups(limit) if strategy == "groups" else g.LimitVectors(limit)
    index = g.Index.grouped(gv, pq, strat)
    rng = np.random.default_rng(9)
    Q = np.concatenate([X[rng.integers(0, n, B - 1)], (rng.standard_normal((1, d)) * 2).astype(np.float32)])
    oi, od, oc = index.batch_query_raw(K, Q)
    ei, ed, ec = oracle.grouped_query(index.data.indices(), d, k, pq.flat_centroids(), cents, offsets, Q, K,
                                      0 if strategy == "groups" else 1, limit)
    assert np.array_equal(oc, ec)
    for q in range(B):
        assert oi[q, :oc[q]].tolist() == ei[q, :ec[q]].tolist(), q
        assert np.array_equal(bits(od[q, :oc[q]]), bits(ed[q, :ec[q]])), q
    index.close()


@pytest.mark.parametrize("dup", [0, 2000])
def test_many_groups_radix_select_of_the_nearest(oracle, g, dup):
    """LimitGroups(70) of ~300 groups: the nearest groups come from the radix-select kernel
    (limit > 63 and limit * 4 <= groups); with duplicated rows centroid distances can tie too."""
    n, d, m, k, B, K = 24000, 16, 4, 16, 11, 10
    X, dm, coarse, gv, pq = _build(oracle, g, n, d, 300, m, k, seed=31, dup=dup, iters=2)
    assert len(gv.centroids) >= 4 * 70
    R, cents, offsets = _oracle_side(oracle, X, coarse, gv, pq, n)
    index = g.Index.grouped(gv, pq, g.LimitGroups(70))
    Q = X[np.random.default_rng(2).integers(0, n, B)]
    oi, od, oc = index.batch_query_raw(K, Q)
    ei, ed, ec = oracle.grouped_query(index.data.indices(), d, k, pq.flat_centroids(), cents, offsets, Q, K, 0, 70)
    assert np.array_equal(oc, ec)
    for q in range(B):
        assert np.array_equal(bits(od[q, :oc[q]]), bits(ed[q, :ec[q]]))
        assert oi[q, :oc[q]].tolist() == ei[q, :ec[q]].tolist()
    index.close()


def test_row0_outside_the_first_cluster_gives_the_reference_empty_group(oracle, g):
    """WordVectors.scala:38-39 seeds the builder with `assignments(0)`: when original row 0 is not in the
    lowest-numbered non-empty cluster the reference emits a leading empty group [0, 0) with a copy of row 0's
    centroid, which LimitGroups(m) then counts as one of the m searched groups."""
    n, d, groups, m, k, K = 5000, 12, 9, 4, 16, 6
    X, dm, coarse, gv, pq = _build(oracle, g, n, d, groups, m, k, seed=77)
    assign = oracle.kmeans_assign(X, 0, d, coarse.centroids, rng_batch=25000)
    if assign[0] == assign.min():                              # make sure the case is the interesting one
        X[[0, int(np.argmax(assign != assign.min()))]] = X[[int(np.argmax(assign != assign.min())), 0]]
        dm = g.DeviceMatrix.from_host(X)
        gv = g.group(dm, coarse)
        pq = g.ProductQuantizer.apply(gv.residuals, g.ProductQuantizerConfig(k, m, 3))
        assign = oracle.kmeans_assign(X, 0, d, coarse.centroids, rng_batch=25000)
    assert assign[0] != assign.min()
    R, cents, offsets = _oracle_side(oracle, X, coarse, gv, pq, n)
    assert gv.offsets[0] == 0 and len(gv.centroids) == len(np.unique(assign)) + 1
    assert np.array_equal(bits(gv.centroids[0]), bits(coarse.centroids[assign[0]]))
    Q = X[[0, 5, 999, 4321]]
    for strat, sid, limit in ((g.LimitGroups(1), 0, 1), (g.LimitGroups(2), 0, 2), (g.LimitVectors(700), 1, 700)):
        index = g.Index.grouped(gv, pq, strat)
        oi, od, oc = index.batch_query_raw(K, Q)
        ei, ed, ec = oracle.grouped_query(index.data.indices(), d, k, pq.flat_centroids(), cents, offsets, Q, K, sid, limit)
        assert np.array_equal(oc, ec)
        for q in range(len(Q)):
            assert oi[q, :oc[q]].tolist() == ei[q, :ec[q]].tolist()
            assert np.array_equal(bits(od[q, :oc[q]]), bits(ed[q, :ec[q]]))
        index.close()


def test_grouping_assigns_in_original_order_then_sorts_by_word(oracle, g):
    """parAssign runs BEFORE the sorts (WordVectors.scala:27-30): with duplicate coarse centroids (init samples
    with replacement) every row draws from the 25 000-row java.util.Random streams, which see the ORIGINAL
    row order, not the word order."""
    rng = np.random.default_rng(3)
    n, d = 60000, 6
    X = (rng.standard_normal((n, d)) + 3.0 * rng.integers(0, 3, (n, 1))).astype(np.float32)
    C0 = X[[5, 5, 900, 900, 17, 5]].copy()                     # duplicates: exact distance ties for every row
    coarse = g.KMeans(d, C0)
    word_order = rng.permutation(n)                            # "sorted by word" is some other order than the rows'
    dm = g.DeviceMatrix.from_host(X)
    gv = g.group(dm, coarse, word_order=word_order)
    assign = oracle.kmeans_assign(X, 0, d, C0, rng_batch=25000)
    assert len(np.unique(assign)) > 3                          # the tie-break really spreads the rows
    perm, cents, offsets = oracle.group_rows(assign, C0, word_order=word_order)
    assert np.array_equal(perm, gv.perm) and np.array_equal(offsets, gv.offsets)
    assert np.array_equal(bits(cents), bits(gv.centroids))
    # assigning the word-sorted rows instead (what this repo did before) gives another grouping
    assign_sorted = oracle.kmeans_assign(X[word_order], 0, d, C0, rng_batch=25000)
    assert not np.array_equal(assign_sorted, assign[word_order])
    R = oracle.group_residuals(X, perm, cents, offsets)
    assert np.array_equal(bits(R), bits(gv.residuals.get_rows(np.arange(n, dtype=np.int32))))
