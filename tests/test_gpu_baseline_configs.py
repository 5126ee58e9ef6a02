"""GPU parity tests AT THE SHAPES of BASELINE.json's configs 3 and 5 (rows scaled down so that the
CPU oracle finishes in seconds; every other dimension is the config's own).

C3: 10 M x 300, KMeans codebook training, m = 32 -> ragged sub-dimensions 12 x 10 + 20 x 9
    (Vectors.scala:84-104), k = 256: the MFMA filter `assign_mfma<5>` with 8 centroid blocks
    (KMeans.scala:24-55), the chain update (KMeans.scala:198-226), ProductQuantizer.apply / encode
    (ProductQuantizer.scala:25-35,121-153).
C5: 10 M x 1024, PQ m = 64 -> s = 16, k = 256: prepareQuery at s = 16 (Index.scala:352-383), the
    2-query-interleaved exact scan (64 KiB of fp32 table per query) and the quantized filter in its
    8-queries-per-entry form (m_pad * 1 KiB <= 144 KiB: m = 64 is eligible, filter.hip filter_eligible) --
    the test asserts through gulon_index_filter_stats that the filter path really ran when it is switched on
    and that the exact scan ran when it is off.  BASELINE's "fp16 distance tables" have no reference
    counterpart: the path here keeps fp32 tables and is bit-exact; no fp16 mode is built.
Everything is compared bit for bit with the oracle."""
import ctypes as C

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gulon_amd
    assert gulon_amd.native.device_count() >= 1
    return gulon_amd


def test_c3_subvector_split(oracle, g):
    fr, un = g.subvector_bounds(300, 32)
    ofr, oun = oracle.subvectors(300, 32)
    assert list(fr) == list(ofr) and list(un) == list(oun)
    w = [u - f for f, u in zip(ofr, oun)]
    assert w == [10] * 12 + [9] * 20


@pytest.mark.parametrize("kind", [2, 3])          # 2 = U[0,1) (BASELINE's C3 data), 3 = overlapping clusters
def test_c3_shape_train_encode_bit_exact(oracle, g, kind):
    n, d, m, k, iters = 20000, 300, 32, 256, 3
    X = oracle.synth(n, d, kind, 1234, 1000)
    dm = g.DeviceMatrix.synthetic(n, d, kind, 1234, 1000)
    pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, iters))
    cents, _, _ = oracle.pq_train(X, m, k, iters)
    assert np.array_equal(bits(pq.flat_centroids()), bits(cents)), "C3-shape codebooks differ"
    enc = pq.encode(dm)
    idx = oracle.pq_encode(X, m, k, cents)
    assert np.array_equal(enc.indices(), idx), "C3-shape PQ codes differ"
    for j in (0, 11, 12, 31):
        assert np.array_equal(enc.encodings[j], oracle.coder_build(8, idx[j]))


@pytest.mark.parametrize("j", [0, 11, 12, 31])     # 10-wide: quantizers 0..11, 9-wide: 12..31
def test_c3_shape_assign_update_one_slice(oracle, g, j):
    """One quantizer's slice: init -> serial assign + parAssign (n > 25 000: the RNG restarts) ->
    fromAssignment, through assign_mfma<5> with nkb = 8."""
    n, d, k = 60000, 300, 256
    X = oracle.synth(n, d, 2, 99, 1000)
    X[40000:40050] = X[:50]                          # duplicated rows: exact ties when they become centroids
    fr, un = oracle.subvectors(d, 32)
    f, s = int(fr[j]), int(un[j] - fr[j])
    dm = g.DeviceMatrix.from_host(X)
    v = g.Vectors(dm, f, f + s)
    km = g.KMeans.init(k, v, j)
    C0, _ = oracle.kmeans_init(X, f, s, k, j)
    assert np.array_equal(bits(km.centroids), bits(C0))
    a = None
    for rb, fn in ((0, km.assign), (25000, km.par_assign)):
        a = fn(v)
        assert np.array_equal(a, oracle.kmeans_assign(X, f, s, C0, rb))
    nxt = g.KMeans.from_assignment(k, s, v, a)
    assert np.array_equal(bits(nxt.centroids), bits(oracle.kmeans_from_assignment(X, f, s, k, a)))


def test_c5_shape_prepare_query(oracle, g):
    d, m, k, B = 1024, 64, 256, 5
    rng = np.random.default_rng(5)
    cents = rng.standard_normal(k * d).astype(np.float32)
    Q = rng.standard_normal((B, d)).astype(np.float32)
    pq = g.ProductQuantizer.from_flat(k, d, m, cents)
    assert np.array_equal(bits(g.prepare_query(pq, Q)), bits(oracle.prepare_query(cents, d, m, k, Q)))


@pytest.mark.parametrize("filt", [1, 0])
def test_c5_shape_train_encode_query_bit_exact(oracle, g, filt):
    n, d, m, k, K, B = 40000, 1024, 64, 256, 10, 24
    X = oracle.synth(n, d, 3, 1234, 100)
    dm = g.DeviceMatrix.synthetic(n, d, 3, 1234, 100)
    pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, 1))
    cents, _, _ = oracle.pq_train(X, m, k, 1)
    assert np.array_equal(bits(pq.flat_centroids()), bits(cents)), "C5-shape codebooks differ"
    index = g.Index.sorted(dm, pq)
    idx = oracle.pq_encode(X, m, k, cents)
    assert np.array_equal(index.vector_index.data.indices(), idx), "C5-shape PQ codes differ"
    rng = np.random.default_rng(11)
    Q = np.concatenate([X[:B // 2], (X[100:100 + B // 2] + 0.05 * rng.standard_normal((B // 2, d))).astype(np.float32)])
    L = g.native.lib()
    g.native.check(L.gulon_index_tuning(index.vector_index._h, b"GULON_SCAN_FILTER", filt))
    try:
        res = index.batch_query(K, Q)
        tiles, redone = C.c_int32(-1), C.c_int32(-1)
        g.native.check(L.gulon_index_filter_stats(index.vector_index._h, C.byref(tiles), C.byref(redone)))
        # 40 000 rows >= the filter's minimum range: switched on, the whole-range query must have taken it
        assert (tiles.value > 0) == bool(filt), (filt, tiles.value)
        sub = index.vector_index.batch_query(K, Q, 777, 33333)
    finally:
        g.native.check(L.gulon_index_tuning(index.vector_index._h, b"GULON_SCAN_FILTER", 1))
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    si, sd, sc = oracle.pq_batch_query(idx, d, k, cents, Q, K, 777, 33333)
    for q in range(B):
        for r, (ei, ed, ec) in ((res[q], (oi, od, oc)), (sub[q], (si, sd, sc))):
            assert len(r) == ec[q]
            assert np.array_equal(bits(r.distances), bits(ed[q, :ec[q]]))
            if r.flags == 0 or (r.flags & 4):
                assert r.rows.tolist() == ei[q, :ec[q]].tolist()
