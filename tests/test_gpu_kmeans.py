"""GPU parity tests for the build path: KMeans init/assign/parAssign/fromAssignment/
computeClusters/iterate and ProductQuantizer train/encode vs the CPU oracle.
Everything is compared bit for bit (centroids as uint32 views, assignments and codes as ints)."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gulon_amd
    assert gulon_amd.native.device_count() >= 1
    return gulon_amd


def _clustered(seed, n, d, kc=8):
    rng = np.random.default_rng(seed)
    cents = rng.uniform(-5, 5, (kc, d))
    scales = rng.uniform(0.1, 1.5, (kc, d))
    which = rng.integers(0, kc, n)
    return (cents[which] + rng.standard_normal((n, d)) * scales[which]).astype(np.float32)


@pytest.mark.parametrize("n,d,frm,s,k,seed", [(3000, 16, 0, 4, 16, 0), (5000, 20, 7, 5, 25, 3), (70000, 8, 0, 8, 256, 1),
                                              (4000, 40, 4, 33, 7, 2), (2000, 140, 3, 130, 5, 4), (500, 3, 1, 1, 3, 9),
                                              # the matrix-core filter's operand layouts: 4 and 5 compact words, three pieces
                                              (30000, 12, 1, 10, 256, 5), (30000, 16, 2, 13, 100, 6), (30000, 16, 1, 14, 256, 7)])
def test_init_assign_update_bit_exact(oracle, g, n, d, frm, s, k, seed):
    X = _clustered(seed, n, d)
    dm = g.DeviceMatrix.from_host(X)
    v = g.Vectors(dm, frm, frm + s)
    km = g.KMeans.init(k, v, seed)
    C0, _ = oracle.kmeans_init(X, frm, s, k, seed)
    assert np.array_equal(bits(km.centroids), bits(C0))
    for rb, fn in ((0, km.assign), (25000, km.par_assign)):
        a = fn(v)
        assert np.array_equal(a, oracle.kmeans_assign(X, frm, s, C0, rb))
    nxt = g.KMeans.from_assignment(k, s, v, a)
    assert np.array_equal(bits(nxt.centroids), bits(oracle.kmeans_from_assignment(X, frm, s, k, a)))


def test_tie_break_rng_across_batches(oracle, g):
    """Duplicate / zero centroids: every row draws from java.util.Random; the stream restarts
    every 25 000 rows in parAssign but not in the serial assign (KMeans.scala:28 vs :71)."""
    rng = np.random.default_rng(5)
    n = 60000
    X = rng.standard_normal((n, 2)).astype(np.float32)
    dm = g.DeviceMatrix.from_host(X)
    v = g.Vectors(dm, 0, 2)
    for C in (np.zeros((4, 2), np.float32),
              np.array([[0.5, 0.5], [0.5, 0.5], [-1, 0], [-1, 0], [0.5, 0.5]], np.float32)):
        km = g.KMeans(2, C)
        for rb, fn in ((0, km.assign), (25000, km.par_assign)):
            a = fn(v)
            exp = oracle.kmeans_assign(X, 0, 2, C, rb)
            assert np.array_equal(a, exp)
            assert len(set(a.tolist())) > 1
    # duplicated data rows => duplicate init centroids (init samples with replacement)
    X2 = np.repeat(rng.standard_normal((50, 3)).astype(np.float32), 40, axis=0)
    dm2 = g.DeviceMatrix.from_host(X2)
    v2 = g.Vectors(dm2, 0, 3)
    km = g.KMeans.init(32, v2, 0)
    C0, _ = oracle.kmeans_init(X2, 0, 3, 32, 0)
    assert np.array_equal(km.par_assign(v2), oracle.kmeans_assign(X2, 0, 3, C0, 25000))
    assert np.array_equal(km.assign(v2), oracle.kmeans_assign(X2, 0, 3, C0, 0))


def test_assign_keeps_untouched_rows_on_nan(oracle, g):
    X = np.ones((100, 2), np.float32)
    X[7] = np.nan
    C = np.array([[0, 0], [1, 1]], np.float32)
    pre = np.full(100, 1, np.int32)
    pre[7] = 0
    a = g.KMeans(2, C).assign(g.Vectors(g.DeviceMatrix.from_host(X), 0, 2), pre.copy())
    exp = oracle.kmeans_assign(X, 0, 2, C, 0, pre.copy())
    assert np.array_equal(a, exp) and a[7] == 0


@pytest.mark.parametrize("n,d,frm,s,k,iters,seed", [(4000, 12, 0, 6, 8, 100, 0), (30000, 16, 8, 8, 64, 5, 2),
                                                    (2000, 4, 0, 4, 300, 3, 1)])
def test_compute_clusters_bit_exact(oracle, g, n, d, frm, s, k, iters, seed):
    X = _clustered(seed + 10, n, d)
    reps = []
    cfg = g.KMeansConfig(k, iters, seed, reps.append)
    km = g.KMeans.compute_clusters(g.Vectors(g.DeviceMatrix.from_host(X), frm, frm + s), cfg)
    Cc, oreps = oracle.kmeans_compute_clusters(X, frm, s, k, iters, seed)
    assert np.array_equal(bits(km.centroids), bits(Cc))
    assert len(reps) == len(oreps)
    for r, o in zip(reps, oreps):
        assert (r.num_iterations, r.converged, r.step_count) == (o["num_iterations"], o["converged"], o["step_count"])
        assert np.float32(r.step_mean).view(np.uint32) == o["step_mean"].view(np.uint32)
        assert np.float32(r.step_s).view(np.uint32) == o["step_s"].view(np.uint32)


def test_iterate_bit_exact_and_descends(oracle, g):
    X = _clustered(21, 3000, 10)
    dm = g.DeviceMatrix.from_host(X)
    v = g.Vectors(dm, 0, 10)
    k0 = g.KMeans.init(6, v)
    cur_o, _ = oracle.kmeans_init(X, 0, 10, 6, 0)
    cur = k0
    for iters in (1, 3, 7):
        cur = cur.iterate(v, iters)
        cur_o = oracle.kmeans_iterate(X, 0, 10, cur_o, iters)
        assert np.array_equal(bits(cur.centroids), bits(cur_o))


def test_degenerate_start_not_stuck(oracle, g):       # KMeansSpec.scala:59-72 on the GPU path
    X = _clustered(33, 1500, 5)
    v = g.Vectors(g.DeviceMatrix.from_host(X), 0, 5)
    a0 = np.zeros(1500, np.int32)
    k0 = g.KMeans.from_assignment(6, 5, v, a0)
    k1 = k0.iterate(v, 1)
    a1 = k1.assign(v)
    e0 = oracle.kmeans_from_assignment(X, 0, 5, 6, a0)
    e1 = oracle.kmeans_iterate(X, 0, 5, e0, 1)
    assert np.array_equal(bits(k1.centroids), bits(e1))
    assert np.array_equal(a1, oracle.kmeans_assign(X, 0, 5, e1, 0))


@pytest.mark.parametrize("n,d,m,k,iters", [(2000, 16, 4, 16, 10), (20000, 32, 4, 256, 4), (3000, 50, 7, 20, 6),
                                           (1500, 10, 10, 3, 5), (1200, 6, 2, 1, 2)])
def test_pq_train_encode_bit_exact(oracle, g, n, d, m, k, iters):
    X = _clustered(n, n, d)
    dm = g.DeviceMatrix.from_host(X)
    pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, iters))
    cents, _, _ = oracle.pq_train(X, m, k, iters)
    assert np.array_equal(bits(pq.flat_centroids()), bits(cents))
    enc = pq.encode(dm)
    idx = oracle.pq_encode(X, m, k, cents)
    assert np.array_equal(enc.indices(), idx)
    w = oracle.coder_width_for_clusters(k)
    for j in range(m):
        assert np.array_equal(enc.encodings[j], oracle.coder_build(w, idx[j]))
    # decode(encode) idempotent (ProductQuantizerSpec.scala:15-26)
    dec = pq.decode(enc)
    assert np.array_equal(bits(dec.data), bits(oracle.pq_decode(idx, d, k, cents)))


def test_index_sorted_end_to_end(oracle, g):
    """Index.sorted + batchQuery: GPU train -> encode -> scan equals the oracle end to end."""
    n, d, m, k = 30000, 64, 8, 256
    X = oracle.synth(n, d, 1, 7, 50)
    dm = g.DeviceMatrix.synthetic(n, d, 1, 7, 50)
    pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, 3))
    si = g.Index.sorted(dm, pq)
    Q = X[[5, 77, 1234, 29999]]
    res = si.batch_query(10, Q)
    cents, _, _ = oracle.pq_train(X, m, k, 3)
    idx = oracle.pq_encode(X, m, k, cents)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, 10)
    for q, r in enumerate(res):
        assert np.array_equal(bits(r.distances), bits(od[q]))
        if r.flags == 0:
            assert r.rows.tolist() == oi[q].tolist()


@pytest.mark.parametrize("n,d,frm,s,k,seed", [
    (20000, 40, 4, 33, 70, 2),        # T = 17 -> streaming kernel with 32 K-steps
    (15000, 128, 0, 128, 300, 4),     # coarse clustering shape: whole 128-d vectors, T = 64
    (30000, 8, 0, 8, 3000, 6),        # short sub-vectors but too many centroids for LDS: T padded to 16
    (9000, 70, 3, 64, 33, 8),         # T = 32, k not a multiple of 32
    (6000, 24, 0, 24, 40, 10),        # duplicated rows and centroids: ties go through the exact path
])
def test_streaming_mfma_assign_bit_exact(oracle, g, n, d, frm, s, k, seed):
    """Long sub-vectors / many centroids use assign_mfma_stream (A operands streamed through LDS)."""
    X = _clustered(seed, n, d)
    if seed == 10:
        X[n // 2:] = X[:n - n // 2]
    dm = g.DeviceMatrix.from_host(X)
    v = g.Vectors(dm, frm, frm + s)
    km = g.KMeans.init(k, v, seed)           # init draws rows with replacement: duplicate centroids happen
    C0, _ = oracle.kmeans_init(X, frm, s, k, seed)
    assert np.array_equal(bits(km.centroids), bits(C0))
    for rb, fn in ((0, km.assign), (25000, km.par_assign)):
        assert np.array_equal(fn(v), oracle.kmeans_assign(X, frm, s, C0, rb))
    it = km.iterate(v, 2)
    assert np.array_equal(bits(it.centroids), bits(oracle.kmeans_iterate(X, frm, s, C0, 2)))


def test_mean_division_identity_holds_for_every_divisor(g):
    """update_chains computes (x - c)/n as q0 = RN(a y), r = fma(-n, q0, a), q = fma(r, y, q0), y = RN(1/n): that must
    be THE correctly rounded quotient (KMeans.scala:218, the JVM's float division) for every n a cluster can reach."""
    import ctypes as C
    bad = C.c_int64(-1)
    assert 0 == g.native.hooks_lib().gulon_selftest_mean_division((1 << 24) - 1, 96, 12345, C.byref(bad))
    assert bad.value == 0


def _stream_update(g, X, frm, s, k, a):
    out = np.zeros((k, s), np.float32)
    X = np.ascontiguousarray(X, np.float32)
    a = np.ascontiguousarray(a, np.int32)
    rc = g.native.hooks_lib().gulon_selftest_stream_update(X.ctypes.data, X.shape[0], X.shape[1], frm, s, k, a.ctypes.data,
                                                           out.ctypes.data)
    assert rc == 0, g.native.hooks_lib().gulon_last_error()
    return out


@pytest.mark.parametrize("n,d,frm,s,k,kind", [
    (100, 4, 0, 4, 7, "plain"),                 # less than one chunk
    (8192, 5, 0, 5, 256, "plain"),              # exactly one chunk, an odd last dimension
    (3 * 8192 + 5, 7, 2, 3, 200, "plain"),      # a ragged last chunk
    (40000, 6, 1, 1, 33, "plain"),              # one dimension (paired with itself)
    (50000, 8, 0, 8, 300, "plain"),             # 257..512 clusters: 512 chain threads
    (50000, 4, 0, 4, 700, "plain"),             # 513..1024 clusters: the chain threads move the chunks themselves
    (30000, 6, 0, 6, 1024, "plain"),
    (40000, 6, 0, 6, 64, "skewed"),             # one cluster holds most rows; some hold none
    (40000, 6, 0, 6, 40, "zeros"),              # sparse rows and duplicates: zero numerators all the way
    (40000, 6, 0, 6, 40, "tiny"),               # denormal and near-denormal values: the plain division's chunks
    (40000, 6, 0, 6, 40, "huge"),               # 2^99 and beyond, infinities, NaNs: every chunk by plain division
])
def test_streamed_update_edge_cases(oracle, g, n, d, frm, s, k, kind):
    """KMeans.fromAssignment (KMeans.scala:198-226) through the streamed update of PQ training: chunk boundaries, the
    three workgroup shapes, and the data the corrected quotient is not trusted with (which must take the plain division
    and still come out bit-identical)."""
    rng = np.random.default_rng(n + k)
    X = rng.standard_normal((n, d)).astype(np.float32)
    a = rng.integers(0, k, n).astype(np.int32)
    if kind == "skewed":
        a = np.where(rng.random(n) < 0.8, 3, rng.integers(0, k // 2, n)).astype(np.int32)
    elif kind == "zeros":
        X[rng.random((n, d)) < 0.7] = 0.0
        X[::3] = X[0]
    elif kind == "tiny":
        m = rng.random((n, d))
        X[m < 0.05] *= np.float32(1e-38)
        X[(m >= 0.05) & (m < 0.1)] *= np.float32(1e-44)
        X[(m >= 0.1) & (m < 0.15)] *= np.float32(1e-33)
    elif kind == "huge":
        m = rng.random((n, d))
        X[m < 0.02] *= np.float32(3e37)
        X[(m >= 0.02) & (m < 0.03)] *= np.float32(1e30)
        X[5000, frm] = np.inf
        X[9000, frm + 1] = -np.inf
        X[12000, frm + 2] = np.nan
    with np.errstate(all="ignore"):
        want = oracle.kmeans_from_assignment(X, frm, s, k, a)
    got = _stream_update(g, X, frm, s, k, a)
    # NaNs compare by position (payloads are not part of the contract: a JVM canonicalises them)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.array_equal(bits(got)[ok], bits(want)[ok])


@pytest.mark.parametrize("n,d,frm,s,k", [(60000, 6, 1, 3, 12000), (80000, 4, 0, 4, 40000)])
def test_more_than_10240_clusters(oracle, g, n, d, frm, s, k):
    """ProductQuantizer.coderFactory allows up to 65 536 clusters per quantizer (ProductQuantizer.scala:11-16):
    beyond the counting sort's LDS counters the stable order of fromAssignment (KMeans.scala:198-226) comes from
    a two-pass radix sort of (row, cluster) pairs."""
    X = _clustered(k, n, d, kc=16)
    dm = g.DeviceMatrix.from_host(X)
    v = g.Vectors(dm, frm, frm + s)
    km = g.KMeans.init(k, v, 3)
    C0, _ = oracle.kmeans_init(X, frm, s, k, 3)
    a = km.par_assign(v)
    assert np.array_equal(a, oracle.kmeans_assign(X, frm, s, C0, 25000))
    nxt = g.KMeans.from_assignment(k, s, v, a)
    assert np.array_equal(bits(nxt.centroids), bits(oracle.kmeans_from_assignment(X, frm, s, k, a)))
    reps = []
    trained = g.KMeans.compute_clusters(v, g.KMeansConfig(k, 2, 5, reps.append))
    Cc, oreps = oracle.kmeans_compute_clusters(X, frm, s, k, 2, 5)
    assert np.array_equal(bits(trained.centroids), bits(Cc)) and len(reps) == len(oreps)


@pytest.mark.parametrize("s", [1, 2, 5, 8, 9, 10, 11, 13, 14, 16])
def test_bf16_split_filter_error_stays_inside_its_band(g, s):
    """The bf16-split MFMA filter (three bf16 pieces per fp32 value, six products -- in 3 / 4 / 5 compact operand words for
    s <= 8 / 10 / 13, one instruction per product beyond) may only differ from the reference's fp32 chain
    (KMeans.scala:42-47) by its proven bound; the band test assumes twice that."""
    import ctypes as C
    worst = 0.0
    for seed, scale in ((1, 1.0), (2, 1e-3), (3, 300.0), (4, 1e6), (5, 1e-12)):
        r = C.c_double(-1.0)
        assert 0 == g.native.hooks_lib().gulon_selftest_assign_band(s, seed, scale, C.byref(r))
        worst = max(worst, r.value)
    assert 0.0 <= worst < 0.5, worst


def test_fused_update_experiment_is_bit_exact():
    """kmeans_fused.hip (KMeans.fromAssignment without the regrouped copy; off by default, GULON_UPDATE_FUSED=1):
    the training loop through it, in a child process (the switch is read once per process), against the oracle --
    ragged sub-dimensions (odd s: the mirrored padding column), k not a multiple of the clusters per workgroup,
    n not a multiple of the chunk, duplicated rows (the plain-division path), a one-chunk problem."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r)
        import gulon_amd as g
        from oracle import oracle
        def bits(a): return np.ascontiguousarray(a, np.float32).view(np.uint32)
        for n, d, m, k, iters, dup in [(30000, 50, 7, 256, 3, 0), (20500, 27, 3, 37, 2, 5000), (900, 12, 4, 16, 2, 0),
                                       (41000, 128, 8, 200, 2, 0)]:
            X = oracle.synth(n, d, 3, 11 + n, 40)
            if dup: X[-dup:] = X[:dup]
            pq = g.ProductQuantizer.apply(g.DeviceMatrix.from_host(X), g.ProductQuantizerConfig(k, m, iters))
            cents, _, _ = oracle.pq_train(X, m, k, iters)
            assert np.array_equal(bits(pq.flat_centroids()), bits(cents)), (n, d, m, k)
        print("fused ok")
    """ % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import gulon_amd
    env = dict(os.environ, GULON_UPDATE_FUSED="1", GULON_HIP_LIB=gulon_amd.native.HOOKS_LIB_PATH)   # the experiment lives in the test-hook build
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "fused ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("n,d,m,k,iters,nbase,part", [(3000, 32, 2, 64, 3, 6, 10), (60000, 48, 3, 80, 2, 8, 10),
                                                     (60000, 48, 3, 80, 2, 8, 150), (1500, 16, 1, 70, 4, 5, 10)])
def test_few_exact_ties_take_the_sparse_replay(oracle, g, n, d, m, k, iters, nbase, part):
    """One row in `part` is a copy of one of a few base rows, so KMeans.init (sampling with replacement) picks some vector
    twice: two identical centroids, and every row nearest to them is an exact tie that draws from java.util.Random --
    with few draws per pass (<= 8192; cases 1, 3, 4) the tie replay takes the stream positions from the drawing rows
    themselves (tie_positions_sparse), across the 25 000-row restarts of parAssign (case 3); case 2 (~20 000 draws per
    sub-quantizer) keeps the dense prefix sums.  Codebooks must equal the oracle's bit for
    bit, which they only do if every draw sits at its place in the stream."""
    rng = np.random.default_rng(n + k)
    X = rng.standard_normal((n, d)).astype(np.float32)
    base = rng.standard_normal((nbase, d)).astype(np.float32)
    copies = rng.permutation(n)[:n // part]
    X[copies] = base[rng.integers(0, nbase, len(copies))]
    pq = g.ProductQuantizer.apply(g.DeviceMatrix.from_host(X), g.ProductQuantizerConfig(k, m, iters))
    cents, _, _ = oracle.pq_train(X, m, k, iters)
    assert np.array_equal(bits(pq.flat_centroids()), bits(cents))
