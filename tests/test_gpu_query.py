"""GPU parity tests for the query path: HIP kernels (through the C ABI) vs the
CPU oracle on the same seeded inputs.  Bit-exact distances and row ids."""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gulon_amd
    assert gulon_amd.native.device_count() >= 1
    return gulon_amd


def _make(oracle, g, n, d, m, k, seed, dup=0):
    rng = np.random.default_rng(seed)
    cents = rng.standard_normal(k * d).astype(np.float32)
    idx = rng.integers(0, k, (m, n)).astype(np.int32)
    if dup:
        idx[:, -dup:] = idx[:, :dup]                      # identical codes => exact distance ties
    pq = g.ProductQuantizer.from_flat(k, d, m, cents)
    coder = pq.coder_factory(n)
    enc = g.EncodedMatrix(coder, [coder.build_code(idx[j]) for j in range(m)])
    return cents, idx, pq, enc


def _same_up_to_ties(rows, dist, orows):
    """An unreplayed tie (IndexSpec.scala:24-32 compares results up to the order inside a tie group): the
    distances are already known to be bit-equal, so the two answers may differ only in WHICH rows of the last
    distance's tie group they hold (a tie that straddles the cut) and in the order inside a group -- every row
    strictly below the last distance must be in both."""
    rows, orows, dist = np.asarray(rows), np.asarray(orows), np.asarray(dist)
    inner = dist < dist[-1] if len(dist) else np.zeros(0, bool)
    assert set(rows[inner].tolist()) == set(orows[inner].tolist())
    assert len(set(rows.tolist())) == len(rows)            # no row twice


def _check(oracle, res, oi, od, oc):
    for q, r in enumerate(res):
        assert len(r) == oc[q]
        assert np.array_equal(bits(r.distances), bits(od[q, :oc[q]]))
        if r.flags == 0 or (r.flags & 4):
            # no tie, or tie resolved by the exact TopKHeap replay: ids and order are the reference's
            assert r.rows.tolist() == oi[q, :oc[q]].tolist()
        else:   # equal distances: order inside a tie group is unspecified (IndexSpec.scala:24-32)
            _same_up_to_ties(r.rows, r.distances, oi[q, :oc[q]])


@pytest.mark.parametrize("d,m,k,B", [(6, 3, 16, 5), (128, 16, 256, 9), (100, 25, 256, 4), (50, 7, 100, 3),
                                     (300, 32, 256, 2), (8, 8, 1, 2), (12, 4, 5, 6)])
def test_prepare_query_bit_exact(oracle, g, d, m, k, B):
    rng = np.random.default_rng(d * 1000 + m)
    cents = rng.standard_normal(k * d).astype(np.float32)
    Q = rng.standard_normal((B, d)).astype(np.float32)
    pq = g.ProductQuantizer.from_flat(k, d, m, cents)
    T = g.prepare_query(pq, Q)
    assert np.array_equal(bits(T), bits(oracle.prepare_query(cents, d, m, k, Q)))


@pytest.mark.parametrize("n,d,m,k,B,K,frm,until", [
    (5000, 6, 3, 16, 5, 10, 0, None),
    (100000, 128, 16, 256, 37, 10, 0, None),        # the BASELINE shape, small n
    (30000, 100, 25, 256, 9, 10, 0, None),          # ragged m (CLI default 25)
    (20000, 64, 32, 256, 6, 5, 0, None),
    (20000, 32, 8, 256, 33, 1, 0, None),
    (9000, 12, 4, 5, 3, 20, 0, None),               # width-4 codes
    (9000, 8, 8, 1, 2, 7, 0, None),                 # width-0 codes (k = 1)
    (9000, 16, 4, 3, 4, 63, 0, None),               # width-2 codes, max K
    (50000, 128, 16, 256, 8, 10, 12345, 40001),     # from/until sub-range
    (1000, 128, 16, 256, 3, 10, 100, 105),          # fewer rows than K
    (1000, 128, 16, 256, 3, 10, 64, 64),            # empty range
    (70, 16, 16, 256, 1, 10, 0, None),
    (20000, 128, 64, 256, 7, 10, 0, None),          # m = 64: 2-query interleave (ds_read_b64)
    (6000, 200, 100, 256, 3, 10, 0, None),          # m = 100: 1-query tables (ds_read_b32)
    (6000, 96, 48, 37, 5, 10, 77, 5000),
])
def test_batch_query_bit_exact(oracle, g, n, d, m, k, B, K, frm, until):
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=n + d)
    Q = np.random.default_rng(7).standard_normal((B, d)).astype(np.float32)
    ix = g.PQIndex(pq, enc)
    res = ix.batch_query(K, Q, frm, until)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K, frm, n if until is None else until)
    _check(oracle, res, oi, od, oc)
    ix.close()


def test_single_query_and_sorted_index(oracle, g):
    n, d, m, k = 40000, 128, 16, 256
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=3)
    q = np.random.default_rng(9).standard_normal(d).astype(np.float32)
    si = g.SortedIndex(g.PQIndex(pq, enc), "cosine")
    r = si.query(10, q)
    qn = oracle.normalize(q)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, qn.reshape(1, -1), 10)
    assert r.rows.tolist() == oi[0].tolist()
    assert np.array_equal(bits(r.distances), bits(od[0]))


def test_ties_are_flagged_and_distances_match(oracle, g):
    n, d, m, k = 6000, 16, 4, 16
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=5, dup=3000)
    Q = np.random.default_rng(1).standard_normal((6, d)).astype(np.float32)
    ix = g.PQIndex(pq, enc)
    res = ix.batch_query(9, Q)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, 9)
    for q, r in enumerate(res):
        assert r.flags & 3                                    # every row has an exact duplicate
        assert r.flags & 4                                    # ... and the heap history was replayed
        assert np.array_equal(bits(r.distances), bits(od[q]))
        assert r.rows.tolist() == oi[q].tolist()              # exactly TopKHeap's ids, in its order
    ix.close()


@pytest.mark.parametrize("n,d,m,k,K,dup,frm,until", [(6000, 16, 4, 16, 9, 3000, 0, None), (40000, 32, 8, 4, 10, 0, 0, None),
                                                    (30000, 16, 16, 2, 63, 0, 5, 29990), (2000, 8, 2, 3, 5, 0, 0, None),
                                                    (100000, 128, 16, 256, 10, 50000, 0, None)])
def test_exact_heap_replay_under_heavy_ties(oracle, g, n, d, m, k, K, dup, frm, until):
    """Few distinct codes => massive exact ties; ids must equal the reference heap's, bit for bit."""
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=n + K, dup=dup)
    Q = np.random.default_rng(K).standard_normal((5, d)).astype(np.float32)
    ix = g.PQIndex(pq, enc)
    res = ix.batch_query(K, Q, frm, until)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K, frm, n if until is None else until)
    for q, r in enumerate(res):
        assert len(r) == oc[q]
        assert np.array_equal(bits(r.distances), bits(od[q, :oc[q]]))
        assert r.flags == 0 or (r.flags & 4), r.flags
        assert r.rows.tolist() == oi[q, :oc[q]].tolist()
    ix.close()


def test_exact_heap_replay_whole_batch_flagged(oracle, g):
    """More flagged queries than one replay wave of 64: every query of the batch is replayed."""
    n, d, m, k, K, B = 20000, 16, 4, 4, 10, 150
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=99)
    Q = np.random.default_rng(3).standard_normal((B, d)).astype(np.float32)
    ix = g.PQIndex(pq, enc)
    oi_g, od_g, oc_g, of_g = ix.batch_query_raw(K, Q)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    assert np.all((of_g & 4) != 0)
    assert np.array_equal(oi_g, oi) and np.array_equal(bits(od_g), bits(od))
    ix.close()


@pytest.mark.parametrize("n,d,m,k,K,B,nbase,frm,until", [
    (300000, 32, 8, 256, 10, 100, 400, 0, None),          # every query tied, m = 8 (4-byte code words)
    (260000, 64, 16, 256, 5, 70, 300, 1000, 255000),      # m = 16 (the headline kernel form), sub-range
    (200000, 32, 8, 256, 63, 40, 100, 0, None),           # largest K of one scan
    (200000, 80, 40, 256, 10, 40, 100, 0, None),          # m = 40: eight queries per table entry, an odd number of query tiles
])
def test_exact_heap_replay_many_flagged_long_range(oracle, g, n, d, m, k, K, B, nbase, frm, until):
    """Enough flagged queries and rows for the long level of the replay to go through the quantized filter
    (replay_level2_filtered): half of the rows are copies of a few hundred code rows => hundreds of exact ties for
    every query; ids and order must be the reference heap's."""
    rng = np.random.default_rng(n + K)
    cents = rng.standard_normal(k * d).astype(np.float32)
    idx = rng.integers(0, k, (m, n)).astype(np.int32)
    base = rng.integers(0, k, (m, nbase)).astype(np.int32)
    copies = rng.permutation(n)[:n // 2]
    idx[:, copies] = base[:, rng.integers(0, nbase, n // 2)]
    # ... and a few rows with ONE far copy each: queries that tie without the bound of the early rows being final
    # (their long level is scanned; the others' is read off the main pass's result, rp_shortcut)
    lone = np.setdiff1d(np.arange(2000, 4000), copies)[:8]
    far = np.setdiff1d(np.arange(n - 3000, n - 1000), copies)[:8]
    idx[:, far] = idx[:, lone]
    pq = g.ProductQuantizer.from_flat(k, d, m, cents)
    coder = pq.coder_factory(n)
    enc = g.EncodedMatrix(coder, [coder.build_code(idx[j]) for j in range(m)])
    ix = g.PQIndex(pq, enc)
    Q = np.stack([ix.decode(int(r)) for r in list(copies[:B - 8]) + list(lone)]).astype(np.float32)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K, frm, n if until is None else until)
    # twice: the handle takes the filtered road once it has SEEN a batch with many flagged queries (a host-mapped
    # hint); the first batch goes through the segment scans, the second through the filter -- same answers
    for _ in range(2):
        oi_g, od_g, oc_g, of_g = ix.batch_query_raw(K, Q, frm, until)
        assert np.array_equal(oc_g, oc) and np.array_equal(bits(od_g), bits(od))
        assert ((of_g & 3) != 0).sum() >= 32                        # enough flagged queries for the filtered level
        replayed = (of_g & 4) != 0
        assert replayed[(of_g & 3) != 0].all()
        assert np.array_equal(oi_g[replayed | (of_g == 0)], oi[replayed | (of_g == 0)])
    ix.close()


@pytest.mark.parametrize("n,d,m,k,B,K,frm,until", [(50000, 64, 16, 256, 5, 64, 0, None), (50000, 64, 16, 256, 3, 100, 0, None),
                                                   (30000, 32, 8, 256, 2, 1000, 100, 29000), (500, 16, 4, 256, 2, 700, 0, None),
                                                   (20000, 40, 10, 256, 9, 127, 0, None), (20000, 40, 10, 256, 2, 128, 0, None)])
def test_large_k_peeling(oracle, g, n, d, m, k, B, K, frm, until):
    """K > 63 (the reference's recall harness asks for up to 1000): peeled 64 entries per scan."""
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=K)
    Q = np.random.default_rng(K).standard_normal((B, d)).astype(np.float32)
    ix = g.PQIndex(pq, enc)
    res = ix.batch_query(K, Q, frm, until)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K, frm, n if until is None else until)
    for q, r in enumerate(res):
        assert len(r) == oc[q]
        assert np.array_equal(bits(r.distances), bits(od[q, :oc[q]]))
        if r.flags == 0:
            assert r.rows.tolist() == oi[q, :oc[q]].tolist()
        else:   # ties beyond the wavefront list keep the (distance, row id) rule: compare as IndexSpec does
            _same_up_to_ties(r.rows, r.distances, oi[q, :oc[q]])
    ix.close()


def test_degenerate_shapes(oracle, g):
    """Empty batch, k_nn = 0, empty index, one row: same (empty) answers as the reference."""
    cents, idx, pq, enc = _make(oracle, g, 300, 16, 4, 16, seed=4)
    ix = g.PQIndex(pq, enc)
    assert ix.batch_query(5, np.zeros((0, 16), np.float32)) == []
    r = ix.batch_query(0, np.zeros((2, 16), np.float32))
    assert [len(x) for x in r] == [0, 0]
    ix.close()
    coder = pq.coder_factory(0)
    empty = g.PQIndex(pq, g.EncodedMatrix(coder, [np.zeros(0, np.uint8)] * 4))
    r = empty.batch_query(3, np.zeros((2, 16), np.float32))
    assert [len(x) for x in r] == [0, 0]
    empty.close()
    coder1 = pq.coder_factory(1)
    one = g.PQIndex(pq, g.EncodedMatrix(coder1, [coder1.build_code(idx[j, :1]) for j in range(4)]))
    Q = np.random.default_rng(0).standard_normal((3, 16)).astype(np.float32)
    res = one.batch_query(4, Q)
    oi, od, oc = oracle.pq_batch_query(np.ascontiguousarray(idx[:, :1]), 16, 16, cents, Q, 4)
    for q, r in enumerate(res):
        assert r.rows.tolist() == oi[q, :oc[q]].tolist() == [0]
        assert np.array_equal(bits(r.distances), bits(od[q, :1]))
    one.close()


def test_concurrent_queries_on_one_index(oracle, g):
    """The recall harness queries one index from many threads (Tests.scala:109-122)."""
    import threading
    n, d, m, k = 20000, 32, 8, 256
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=21)
    ix = g.PQIndex(pq, enc)
    Q = np.random.default_rng(2).standard_normal((16, d)).astype(np.float32)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, 10)
    errs = []

    def work(q):
        try:
            for _ in range(5):
                r = ix.query(10, Q[q])
                assert r.rows.tolist() == oi[q].tolist()
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(q,)) for q in range(16)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    ix.close()


def test_requirements_raise(oracle, g):
    cents, idx, pq, enc = _make(oracle, g, 1000, 16, 4, 16, seed=2)
    ix = g.PQIndex(pq, enc)
    Q = np.zeros((1, 16), np.float32)
    with pytest.raises(ValueError):
        ix.batch_query(5, Q, 10, 5)          # require(from <= until)   Index.scala:418
    with pytest.raises(ValueError):
        ix.batch_query(5, Q, 0, 1001)        # require(until <= length) Index.scala:419
    with pytest.raises(ValueError):
        ix.batch_query(5, Q, -1, 10)
    with pytest.raises(NotImplementedError):
        ix.batch_query(9000, Q)              # above GULON_MAX_K_PEELED: loud, not silent
    ix.close()


def test_sharded_partials_merge_equals_full(oracle, g):
    """Row-sharded scan + TopKHeap.merge semantics == unsharded scan (multi-GPU path on one GPU)."""
    import ctypes as C
    n, d, m, k, B, K = 30000, 64, 16, 256, 11, 10
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=11)
    Q = np.random.default_rng(4).standard_normal((B, d)).astype(np.float32)
    full = g.PQIndex(pq, enc).batch_query(K, Q)
    bounds = [0, 7000, 7001, 19999, n]
    pd, pi = [], []
    N = g.native
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        coder = pq.coder_factory(hi - lo)
        sub = g.EncodedMatrix(coder, [coder.build_code(idx[j, lo:hi]) for j in range(m)])
        shard = g.PQIndex(pq, sub, row_base=lo)
        dq, dv, di = C.c_void_p(), C.c_void_p(), C.c_void_p()
        N.check(N.lib().gulon_dev_malloc(C.byref(dq), Q.nbytes))
        N.check(N.lib().gulon_dev_malloc(C.byref(dv), B * (K + 1) * 4))
        N.check(N.lib().gulon_dev_malloc(C.byref(di), B * (K + 1) * 4))
        N.check(N.lib().gulon_memcpy_h2d(dq, Q.ctypes.data_as(C.c_void_p), Q.nbytes))
        N.check(N.lib().gulon_index_scan_partial_dev(shard._h, dq, B, K, 0, hi - lo, dv, di, None))
        N.check(N.lib().gulon_device_synchronize())
        v = np.zeros((B, K + 1), np.float32)
        i = np.zeros((B, K + 1), np.int32)
        N.check(N.lib().gulon_memcpy_d2h(v.ctypes.data_as(C.c_void_p), dv, v.nbytes))
        N.check(N.lib().gulon_memcpy_d2h(i.ctypes.data_as(C.c_void_p), di, i.nbytes))
        for p in (dq, dv, di):
            N.check(N.lib().gulon_dev_free(p))
        pd.append(v)
        pi.append(i)
        shard.close()
    from gulon_amd.topk import merge_partials
    oi, od, oc, of = merge_partials(np.stack(pd), np.stack(pi), K)
    for q in range(B):
        assert oi[q].tolist() == full[q].rows.tolist()
        assert np.array_equal(bits(od[q]), bits(full[q].distances))


def test_recall_harness_matches_oracle(oracle, g):
    """Tests.recallOf (Tests.scala:18-41) through the GPU path == the oracle's restatement."""
    from gulon_amd.recall import recall_at_k, sample_rows
    n, d, m, k = 20000, 32, 8, 256
    X = oracle.synth(n, d, 3, 5, 40)
    dm = g.DeviceMatrix.from_host(X)
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=8)
    rows = sample_rows(n, 50, 0)
    r0 = oracle.JavaRandom(0)
    assert rows.tolist() == [r0.next_int(n) for _ in range(50)]          # Tests.sample: Random(0).nextInt(n)
    Q = X[rows]
    ix = g.PQIndex(pq, enc)
    for K in (10, 100):
        oi, od, oc, of = ix.batch_query_raw(K, Q)
        mean, sd = recall_at_k(dm, Q, K, oi, oc)
        ei, ed, ec = oracle.exact_knn(X, Q, K)
        omean, osd = oracle.recall(X, Q, K, oi, oc, ed, ec)
        assert np.float32(mean) == np.float32(omean) and abs(sd - osd) < 1e-6
    ix.close()


def test_synthetic_data_matches_oracle(oracle, g):
    for kind in (0, 1, 2, 3):
        dm = g.DeviceMatrix.synthetic(3000, 50, kind, 1234, 17)
        X = dm.to_host()
        assert np.array_equal(bits(X), bits(oracle.synth(3000, 50, kind, 1234, 17)))


@pytest.mark.parametrize("n,d,B,K,frm,until", [(100000, 50, 40, 10, 0, None), (5000, 128, 3, 63, 0, None),
                                                (5000, 7, 17, 5, 123, 4000), (300, 33, 2, 10, 0, None),
                                                (20000, 24, 3, 64, 0, None), (20000, 24, 2, 1000, 50, 19000),
                                                (400, 5, 2, 500, 0, None)])
def test_exact_knn_bit_exact(oracle, g, n, d, B, K, frm, until):
    X = oracle.synth(n, d, 0, 99)
    Q = oracle.synth(B, d, 0, 100)
    res = g.exact_nearest_neighbours(g.DeviceMatrix.from_host(X), Q, K, frm, until)
    oi, od, oc = oracle.exact_knn(X, Q, K, frm, until)
    _check(oracle, res, oi, od, oc)
