"""More than 256 centroids per quantizer: the 10/12/16-bit codes of Coder.BytePlus
(Coder.scala:99-127,142-168; ProductQuantizer.coderFactory) through the wide-code path
(wide.hip) -- tables, distances and neighbour ids against the CPU oracle, bit for bit."""
import numpy as np
import pytest

from conftest import bits
from test_gpu_query import _check, _make

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gulon_amd
    assert gulon_amd.native.device_count() >= 1
    return gulon_amd


@pytest.mark.parametrize("d,m,k,B", [(16, 4, 257, 5), (64, 16, 1024, 3), (30, 7, 1000, 4), (12, 3, 4097, 2)])
def test_prepare_query_wide_bit_exact(oracle, g, d, m, k, B):
    rng = np.random.default_rng(k)
    cents = rng.standard_normal(k * d).astype(np.float32)
    Q = rng.standard_normal((B, d)).astype(np.float32)
    pq = g.ProductQuantizer.from_flat(k, d, m, cents)
    assert np.array_equal(bits(g.prepare_query(pq, Q)), bits(oracle.prepare_query(cents, d, m, k, Q)))


@pytest.mark.parametrize("n,d,m,k,B,K,frm,until", [
    (20000, 32, 8, 257, 9, 10, 0, None),            # width 10, table in LDS
    (30000, 64, 16, 1024, 7, 10, 0, None),          # width 10, 64 KiB table
    (10000, 16, 4, 4096, 5, 5, 0, None),            # width 12
    (8000, 48, 12, 5000, 4, 10, 0, None),           # width 16, table gathered from memory (240 KB)
    (3000, 12, 3, 65536, 3, 3, 0, None),            # the largest code book: one quantizer's entries exceed LDS
    (6000, 27, 9, 10000, 5, 10, 0, None),           # 40 KB per quantizer: three slices of three (first, middle, last)
    (9000, 16, 8, 6000, 4, 7, 777, 8100),           # sliced table over a sub-range
    (25000, 30, 7, 300, 6, 1, 1234, 20001),         # ragged m, sub-range with partial first and last block
    (9000, 20, 5, 777, 70, 63, 0, None),            # largest K, ragged batch
    (40, 8, 2, 300, 3, 10, 0, None),                # fewer rows than one block; K <= n
    (7, 8, 2, 300, 2, 10, 0, None),                 # fewer rows than K
])
def test_wide_query_bit_exact(oracle, g, n, d, m, k, B, K, frm, until):
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=n + k)
    assert enc.coder.width in (10, 12, 16)
    Q = np.random.default_rng(5).standard_normal((B, d)).astype(np.float32)
    ix = g.PQIndex(pq, enc)
    res = ix.batch_query(K, Q, frm, until)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K, frm, n if until is None else until)
    _check(oracle, res, oi, od, oc)
    ix.close()


@pytest.mark.parametrize("n,d,m,k,B,K,dup", [(12000, 16, 4, 1000, 6, 10, 2000), (150000, 32, 8, 4096, 40, 7, 60000),
                                            (9000, 8, 2, 40000, 5, 5, 3000)])
def test_wide_ties_are_replayed_exactly(oracle, g, n, d, m, k, B, K, dup):
    """Duplicate rows => exact distance ties => the literal TopKHeap replay, also over 16-bit codes: ids and order
    are the reference heap's (TopKHeap.scala:57-79)."""
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=3, dup=dup)
    ix = g.PQIndex(pq, enc)
    Q = np.stack([ix.decode(r) for r in range(0, dup, dup // B)][:B]).astype(np.float32)
    oi, od, oc, of = ix.batch_query_raw(K, Q)
    ei, ed, ec = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    assert np.array_equal(bits(od), bits(ed)) and np.array_equal(oc, ec)
    assert ((of & 3) != 0).all() and ((of & 4) != 0).all()     # tie flags, and every one of them replayed
    assert np.array_equal(oi, ei)
    ix.close()


def test_wide_sharded_equals_unsharded(g):
    from test_gpu_shared_bounds import _same, sharded_query
    n, d, m, k, B, K = 90000, 32, 8, 600, 20, 10
    rng = np.random.default_rng(8)
    cents = rng.standard_normal(k * d).astype(np.float32)
    idx = rng.integers(0, k, (m, n)).astype(np.int32)
    pq = g.ProductQuantizer.from_flat(k, d, m, cents)
    coder = pq.coder_factory(n)
    enc = g.EncodedMatrix(coder, [coder.build_code(idx[j]) for j in range(m)])
    Q = rng.standard_normal((B, d)).astype(np.float32)
    full = g.PQIndex(pq, enc).batch_query_raw(K, Q)
    _same(sharded_query(g, pq, enc, n, 3, Q, K), full)


def test_pq_train_encode_query_wide_end_to_end(oracle, g):
    """ProductQuantizer.apply with numClusters = 300 (width-10 codes) -> encode -> index -> query."""
    n, d, m, k, iters, B, K = 4000, 12, 3, 300, 3, 8, 5
    X = oracle.synth(n, d, 3, 11, 40)
    dm = g.DeviceMatrix.from_host(X)
    pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, iters))
    cents, _, _ = oracle.pq_train(X, m, k, iters)
    assert np.array_equal(bits(pq.flat_centroids()), bits(cents))
    enc = pq.encode(dm)
    idx = oracle.pq_encode(X, m, k, cents)
    assert np.array_equal(enc.indices(), idx)
    for j in range(m):
        assert np.array_equal(enc.encodings[j], oracle.coder_build(10, idx[j]))
    Q = X[::n // B][:B]
    res = g.PQIndex(pq, enc).batch_query(K, Q)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    _check(oracle, res, oi, od, oc)


@pytest.mark.parametrize("k,strategy,limit", [(300, "groups", 2), (1024, "groups", 5), (300, "vectors", 700), (5000, "groups", 3)])
def test_grouped_index_over_wide_codes(oracle, g, k, strategy, limit):
    """GroupedIndex with more than 256 centroids per residual quantizer (Coder.BytePlus codes): per-group literal
    TopKHeaps over 16-bit codes, folded with TopKHeap.merge -- ids, order and distances equal the oracle's."""
    n, d, m, groups, B, K = 6000, 8, 2, 6, 7, 6
    rng = np.random.default_rng(k)
    X = (rng.standard_normal((n, d)) + 3.0 * rng.integers(0, 3, (n, 1))).astype(np.float32)
    X[-500:] = X[:500]                                           # ties inside and across groups
    dm = g.DeviceMatrix.from_host(X)
    coarse = g.KMeans.compute_clusters(g.Vectors(dm), g.KMeansConfig(groups, 2))
    gv = g.group(dm, coarse)
    pq = g.ProductQuantizer.apply(gv.residuals, g.ProductQuantizerConfig(k, m, 1))
    strat = g.LimitGroups(limit) if strategy == "groups" else g.LimitVectors(limit)
    index = g.Index.grouped(gv, pq, strat)
    Q = np.concatenate([X[rng.integers(0, n, B - 1)], rng.standard_normal((1, d)).astype(np.float32)])
    oi, od, oc = index.batch_query_raw(K, Q)
    ei, ed, ec = oracle.grouped_query(index.data.indices(), d, k, pq.flat_centroids(), gv.centroids, gv.offsets, Q, K,
                                      0 if strategy == "groups" else 1, limit)
    assert np.array_equal(oc, ec)
    for q in range(B):
        assert oi[q, :oc[q]].tolist() == ei[q, :ec[q]].tolist()
        assert np.array_equal(bits(od[q, :oc[q]]), bits(ed[q, :ec[q]]))
    index.close()


def _filtered(g, ix):
    """query tiles of the last batch that went through the 8-bit filter, and of those redone by the exact scan"""
    import ctypes as C
    from gulon_amd import native as N
    t, r = C.c_int32(-1), C.c_int32(-1)
    N.check(N.lib().gulon_index_filter_stats(ix._h, C.byref(t), C.byref(r)))
    return t.value, r.value


@pytest.mark.parametrize("n,d,m,k,B,K,frm,until", [
    (40000, 64, 16, 1024, 21, 10, 0, None),         # 8 queries per entry (128 KiB of 8-bit tables)
    (30000, 32, 8, 257, 9, 1, 0, None),             # smallest wide code book, K = 1
    (25000, 30, 7, 300, 6, 10, 1234, 20001),        # ragged m (a last group of three quantizers), sub-range
    (50000, 32, 8, 4096, 13, 63, 0, None),          # 4 queries per entry (128 KiB), largest K
    (20000, 20, 5, 777, 70, 5, 64, 19999),          # more queries than one launch row, odd k
    (9000, 16, 16, 2048, 5, 10, 0, None),           # 4 queries per entry at m = 16
    (30000, 32, 16, 4096, 11, 10, 0, None),         # the table in two slices of eight quantizers, byte sums parked
    (20000, 24, 12, 5000, 7, 10, 100, 19000),       # odd k: slices of 7 + 5 quantizers, sub-range
    (12000, 16, 8, 16384, 6, 3, 0, None),           # two quantizers per slice: four slices
    (10000, 8, 4, 32768, 3, 10, 0, None),           # one quantizer per slice (the largest sliced code book)
])
def test_wide_filter_bit_exact(oracle, g, n, d, m, k, B, K, frm, until):
    """The quantized lower-bound filter over 16-bit codes (wide_filter.hip): same ids, order and distance bits as the
    oracle (and therefore as the exact wide scan), on clustered data where the bounds do prune."""
    import ctypes as C
    from gulon_amd import native as N
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=n + k)
    ix = g.PQIndex(pq, enc)
    N.check(N.lib().gulon_index_tuning(ix._h, b"GULON_FILTER_MIN_RB", 4))     # small ranges through the filter too
    rng = np.random.default_rng(5)
    Q = np.concatenate([np.stack([ix.decode(int(r)) for r in rng.integers(0, n, B - 2)]),
                        rng.standard_normal((2, d))]).astype(np.float32)
    res = ix.batch_query(K, Q, frm, until)
    tiles, redone = _filtered(g, ix)
    assert tiles == B                                            # the batch took the filter ...
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K, frm, n if until is None else until)
    _check(oracle, res, oi, od, oc)
    # ... and equals the exact wide scan of the same index
    N.check(N.lib().gulon_index_tuning(ix._h, b"GULON_SCAN_FILTER", 0))
    res2 = ix.batch_query(K, Q, frm, until)
    assert _filtered(g, ix)[0] == 0
    for a, b in zip(res, res2):
        assert a.rows.tolist() == b.rows.tolist() and np.array_equal(bits(a.distances), bits(b.distances))
    ix.close()


def test_wide_filter_unusable_bounds_fall_back(oracle, g):
    """NaN / huge queries (no finite bound) and a tiny survivor queue (every sub-queue overflows): those queries are
    redone by the exact scan on the device -- results unchanged."""
    import ctypes as C
    from gulon_amd import native as N
    n, d, m, k, B, K = 30000, 32, 8, 1024, 12, 10
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=99)
    ix = g.PQIndex(pq, enc)
    N.check(N.lib().gulon_index_tuning(ix._h, b"GULON_FILTER_MIN_RB", 4))
    N.check(N.lib().gulon_index_tuning(ix._h, b"GULON_FILTER_CAP", 64))      # 4 entries per sub-queue... clamped to 64
    rng = np.random.default_rng(1)
    Q = rng.standard_normal((B, d)).astype(np.float32)                       # random queries: loose bounds, many survivors
    Q[3, 5] = np.float32(3e19)                                               # every distance +inf
    oi, od, oc, of = ix.batch_query_raw(K, Q)
    tiles, redone = _filtered(g, ix)
    assert tiles == B and redone >= 1
    ei, ed, ec = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    for q in range(B):
        if q == 3:
            continue                                                         # (non-finite distances: the (distance, row) rule on wide indexes)
        assert np.array_equal(bits(od[q]), bits(ed[q])) and (of[q] & 3 or oi[q].tolist() == ei[q].tolist())
    ix.close()


@pytest.mark.parametrize("n,d,m,k,B,K", [(90000, 32, 8, 600, 20, 10),
                                         (60000, 32, 16, 4096, 9, 10)])      # the 8-bit tables in two slices
def test_wide_filter_sharded_equals_unsharded(g, n, d, m, k, B, K):
    from test_gpu_shared_bounds import _same, sharded_query
    import ctypes as C
    from gulon_amd import native as N
    import os
    before = os.environ.get("GULON_FILTER_MIN_RB")
    g.tune_live(GULON_FILTER_MIN_RB=4)
    try:
        rng = np.random.default_rng(8)
        cents = rng.standard_normal(k * d).astype(np.float32)
        idx = rng.integers(0, k, (m, n)).astype(np.int32)
        pq = g.ProductQuantizer.from_flat(k, d, m, cents)
        coder = pq.coder_factory(n)
        enc = g.EncodedMatrix(coder, [coder.build_code(idx[j]) for j in range(m)])
        Q = rng.standard_normal((B, d)).astype(np.float32)
        full = g.PQIndex(pq, enc).batch_query_raw(K, Q)
        _same(sharded_query(g, pq, enc, n, 3, Q, K), full)
    finally:
        g.tune_live(GULON_FILTER_MIN_RB=512)
        if before is None:
            os.environ.pop("GULON_FILTER_MIN_RB", None)
        else:
            os.environ["GULON_FILTER_MIN_RB"] = before
