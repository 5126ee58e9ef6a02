"""The reference's own ScalaCheck properties, ported and
run against the C oracle.  File:line cite /root/reference/core/src/test/scala/net/tixxit/gulon/."""
import numpy as np
from hypothesis import given, settings, strategies as st, HealthCheck

from conftest import bits

SET = dict(deadline=None, max_examples=60, suppress_health_check=[HealthCheck.too_slow])
f32s = st.floats(width=32, allow_nan=False, allow_infinity=False)


@settings(**SET)
@given(st.lists(st.tuples(st.integers(-2**31, 2**31 - 1), f32s), max_size=60), st.data())
def test_topk_heap_first_k(oracle, kvs, data):                       # TopKHeapSpec.scala:16-31
    k = data.draw(st.integers(1, max(2 * len(kvs), 1)))
    h = oracle.TopKHeap(k)
    for i, x in kvs:
        h.update(i, x)
    ks, vs = h.drain()
    exp = sorted(kvs, key=lambda kv: np.float32(kv[1]))[:k]          # stable sortBy(_._2).take(k)
    assert sorted(np.float32(v) for v in vs) == [np.float32(v) for _, v in exp]
    assert vs.tolist() == sorted(vs.tolist())
    if len({np.float32(v) for _, v in kvs}) == len(kvs):             # tie-free => exact equality
        assert list(zip(ks.tolist(), vs.tolist())) == [(i, float(np.float32(v))) for i, v in exp]


@settings(**SET)
@given(st.lists(st.lists(st.tuples(st.integers(0, 10**6), f32s), max_size=20), max_size=6), st.data())
def test_topk_heap_merge(oracle, groups, data):                      # TopKHeapSpec.scala:33-52
    total = sum(len(g) for g in groups)
    k = data.draw(st.integers(1, max(2 * total, 1)))
    h = oracle.TopKHeap(k)
    for g in groups:
        h0 = oracle.TopKHeap(k)
        for i, x in g:
            h0.update(i, x)
        h.merge(h0)
    _, vs = h.drain()
    exp = sorted(np.float32(v) for g in groups for _, v in g)[:k]
    assert vs.tolist() == [float(v) for v in exp]


@settings(**SET)
@given(st.integers(1, 16), st.data())
def test_coder_round_trip(oracle, width, data):                      # CoderSpec.scala:17-27
    vals = data.draw(st.lists(st.integers(0, (1 << width) - 1), min_size=1, max_size=40))
    w = oracle.coder_round_width(width)
    code = oracle.coder_build(w, vals)
    assert code.size == oracle.coder_bytes(w, len(vals))
    assert [oracle.coder_get(w, code, len(vals), i) for i in range(len(vals))] == vals


def _clustered(rng, d, k, per):
    cents = rng.uniform(-5, 5, (k, d))
    scales = rng.uniform(-5, 5, (k, d))
    pts = [cents[c] + rng.standard_normal((max(per, 1), d)) * scales[c] for c in range(k)]
    return np.concatenate(pts).astype(np.float32), cents.astype(np.float32)


def _objective(oracle, X, C):
    a = oracle.kmeans_assign(X, 0, X.shape[1], C, 0)
    return sum(float(oracle.distance_sq(X[i], C[a[i]])) for i in range(X.shape[0]))


@settings(deadline=None, max_examples=25)
@given(st.integers(0, 2**31 - 1), st.integers(2, 12), st.integers(2, 8), st.integers(5, 20))
def test_kmeans_converges_and_descends(oracle, seed, d, k, per):     # KMeansSpec.scala:23-57
    X, _ = _clustered(np.random.default_rng(seed), d, k, per)
    C, reps = oracle.kmeans_compute_clusters(X, 0, d, k, 100, 0)
    assert reps[-1]["converged"]
    C0, _ = oracle.kmeans_init(X, 0, d, k, 0)
    o = _objective(oracle, X, C0)
    cur = C0
    for iters in (1, 3, 7, 11):
        cur = oracle.kmeans_iterate(X, 0, d, cur, iters)
        o2 = _objective(oracle, X, cur)
        assert o >= o2 or abs(o - o2) <= 1e-4 * abs(o)
        o = o2


@settings(deadline=None, max_examples=25)
@given(st.integers(0, 2**31 - 1), st.integers(2, 8), st.integers(2, 8))
def test_kmeans_not_stuck_on_duplicates(oracle, seed, d, k):         # KMeansSpec.scala:59-72
    X, _ = _clustered(np.random.default_rng(seed), d, k, 8)
    a0 = np.zeros(X.shape[0], np.int32)
    k0 = oracle.kmeans_from_assignment(X, 0, d, k, a0)
    k1 = oracle.kmeans_iterate(X, 0, d, k0, 1)
    a1 = oracle.kmeans_assign(X, 0, d, k1, 0)
    if not np.array_equal(a0, a1):
        assert _objective(oracle, X, k0) > _objective(oracle, X, k1)


@settings(deadline=None, max_examples=30)
@given(st.integers(0, 2**31 - 1), st.integers(1, 5), st.integers(1, 4), st.integers(1, 100), st.integers(1, 30))
def test_pq_encode_decode(oracle, seed, m, sd, k, n):                # ProductQuantizerSpec.scala:15-68
    rng = np.random.default_rng(seed)
    d = m * sd
    cents = rng.uniform(-1, 1, k * d).astype(np.float32)
    X = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    idx = oracle.pq_encode(X, m, k, cents)
    dec = oracle.pq_decode(idx, d, k, cents)
    idx2 = oracle.pq_encode(dec, m, k, cents)
    dec2 = oracle.pq_decode(idx2, d, k, cents)
    assert np.allclose(dec, dec2, rtol=1e-3, atol=0)                 # idempotent within 1e-3
    # decode of codes 0..k-1 returns exactly the centroids
    allc = np.tile(np.arange(k, dtype=np.int32), (m, 1))
    back = oracle.pq_decode(allc, d, k, cents)
    fr, un = oracle.subvectors(d, m)
    for j in range(m):
        cb = cents[k * fr[j]: k * un[j]].reshape(k, un[j] - fr[j])
        assert np.array_equal(bits(back[:, fr[j]:un[j]]), bits(cb))
    # encode picks a code at least as close as random codes
    p = X[0]
    best = np.sqrt(float(oracle.distance_sq(p, dec[0])))
    rnd = oracle.pq_decode(rng.integers(0, k, (m, 8)).astype(np.int32), d, k, cents)
    for r in rnd:
        assert best <= np.sqrt(float(oracle.distance_sq(p, r))) * (1 + 1e-6) + 1e-7


@settings(deadline=None, max_examples=30)
@given(st.integers(0, 2**31 - 1), st.integers(1, 5), st.integers(1, 4), st.integers(1, 100), st.integers(1, 60))
def test_index_query_equals_exact_over_decoded(oracle, seed, m, sd, k, n):   # IndexSpec.scala:24-43
    rng = np.random.default_rng(seed)
    d = m * sd
    cents = rng.uniform(-1, 1, k * d).astype(np.float32)
    X = rng.uniform(-1, 1, (n, d)).astype(np.float32)
    q = rng.uniform(-1, 1, (1, d)).astype(np.float32)
    K = int(rng.integers(1, n + 1))
    idx = oracle.pq_encode(X, m, k, cents)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, q, K)
    dec = oracle.pq_decode(idx, d, k, cents)
    ei, ed, ec = oracle.exact_knn(dec, q, K)
    assert oc[0] == ec[0] == K
    # assertResultsMatch: sort actual by (distSq, expected rank), compare ids
    rank = {r: i for i, r in enumerate(ei[0].tolist())}
    reordered = [r for _, _, r in sorted((float(od[0, i]), rank.get(int(oi[0, i]), 0), int(oi[0, i]))
                                         for i in range(K))]
    exact_d = {int(ei[0, i]): float(ed[0, i]) for i in range(K)}
    # ADC sums and direct sums round differently: allow near-ties to reorder (1e-4 rel, TestUtils.scala:4-8)
    for a, b in zip(reordered, ei[0].tolist()):
        if a != b:
            da = float(oracle.distance_sq(q[0], dec[a]))
            assert abs(da - exact_d[b]) <= 1e-4 * max(abs(da), abs(exact_d[b])) + 1e-12
