"""GPU tests of (1) query contexts -- several batches in flight over ONE copy of the codes -- and
(2) the C ABI's own multi-GPU index (gulon_sharded_index_*: one process, RCCL all-gathers), both against
the CPU oracle and the unsharded index.  PQIndex.batchQuery: Index.scala:417-440; TopKHeap.merge:
TopKHeap.scala:44-53; literal heap under ties: TopKHeap.scala:57-79; concurrent callers: Tests.scala:109-122."""
import ctypes as C
import threading

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gulon_amd
    assert gulon_amd.native.device_count() >= 1
    return gulon_amd


def _make(g, n, d, m, k, seed, dup=0):
    rng = np.random.default_rng(seed)
    cents = rng.standard_normal(k * d).astype(np.float32)
    idx = rng.integers(0, k, (m, n)).astype(np.int32)
    if dup:
        idx[:, -dup:] = idx[:, :dup]
    pq = g.ProductQuantizer.from_flat(k, d, m, cents)
    coder = pq.coder_factory(n)
    enc = g.EncodedMatrix(coder, [coder.build_code(idx[j]) for j in range(m)])
    return cents, idx, pq, enc


def _check(res, oi, od, oc):
    ri, rd, rc, rf = res
    assert np.array_equal(rc, oc)
    for q in range(len(rc)):
        assert np.array_equal(bits(rd[q, :oc[q]]), bits(od[q, :oc[q]]))
        if rf[q] == 0 or (rf[q] & 4):
            assert ri[q, :oc[q]].tolist() == oi[q, :oc[q]].tolist()


def _same(a, b):
    assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])


@pytest.mark.parametrize("n", [90000, 20000])       # filtered scan / exact scan
def test_two_streams_one_handle_and_two_contexts(oracle, g, n):
    """Batches in flight on different streams: on ONE handle they are ordered on the device (never corrupt
    each other's scratch); on two contexts of one index they overlap.  Different queries per batch."""
    import torch
    from gulon_amd import native as N
    d, m, k, B, K = 64, 16, 256, 40, 10
    cents, idx, pq, enc = _make(g, n, d, m, k, seed=n)
    rng = np.random.default_rng(1)
    Qs = [rng.standard_normal((B, d)).astype(np.float32) for _ in range(4)]
    exp = [oracle.pq_batch_query(idx, d, k, cents, Q, K) for Q in Qs]
    ix = g.PQIndex(pq, enc)
    ctx = ix.context()
    dev = torch.device("cuda", 0)
    L = N.lib()
    streams = [torch.cuda.Stream() for _ in range(4)]
    qd = [torch.from_numpy(Q).to(dev) for Q in Qs]
    outs = [(torch.empty((B, K), dtype=torch.int32, device=dev), torch.empty((B, K), dtype=torch.float32, device=dev),
             torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev)) for _ in Qs]
    torch.cuda.synchronize()
    for rep in range(3):
        for i in range(4):
            h = ix._h if i % 2 == 0 else ctx._h        # batches 0, 2 on the index itself; 1, 3 on its context
            oi, od, oc, of = outs[i]
            N.check(L.gulon_index_batch_query_dev(h, qd[i].data_ptr(), B, K, 0, n, oi.data_ptr(), od.data_ptr(),
                                                  oc.data_ptr(), of.data_ptr(), C.c_void_p(streams[i].cuda_stream)))
    torch.cuda.synchronize()
    for i in range(4):
        _check(tuple(t.cpu().numpy() for t in outs[i]), *exp[i])
    ctx.close()
    # the codes outlive the index while a context is alive
    ctx2 = ix.context()
    ix.close()
    _check(ctx2.batch_query_raw(K, Qs[0]), *exp[0])
    ctx2.close()


def test_concurrent_host_queries_from_threads(oracle, g):
    """Tests.recallOf queries one index from a thread pool (Tests.scala:109-122): the host-pointer entry point
    gives every concurrent caller a workspace of its own."""
    n, d, m, k, K = 70000, 32, 8, 256, 10
    cents, idx, pq, enc = _make(g, n, d, m, k, seed=4)
    ix = g.PQIndex(pq, enc)
    rng = np.random.default_rng(2)
    Qs = [rng.standard_normal((7 + t, d)).astype(np.float32) for t in range(8)]
    res, err = [None] * 8, []

    def run(t):
        try:
            for _ in range(3):
                res[t] = ix.batch_query_raw(K, Qs[t])
        except Exception as e:      # pragma: no cover
            err.append(e)

    ts = [threading.Thread(target=run, args=(t,)) for t in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not err, err
    for t in range(8):
        _check(res[t], *oracle.pq_batch_query(idx, d, k, cents, Qs[t], K))
    ix.close()


def test_bounded_scan_dropped_by_an_interleaved_query(g):
    import torch
    from gulon_amd.sharded import HipEngine
    n, d, m, k, B, K = 100000, 32, 8, 256, 4, 3
    cents, idx, pq, enc = _make(g, n, d, m, k, seed=6)
    eng = HipEngine(pq, enc, 0, torch.device("cuda", 0))
    q = eng.to_device(np.random.default_rng(0).standard_normal((B, d)).astype(np.float32))
    bd = eng.alloc((B, K + 1), "f32")
    pv, pi = eng.alloc((B, K + 1), "f32"), eng.alloc((B, K + 1), "i32")
    eng.scan_bounds(q, B, K, bd)
    eng.scan_partial(q, B, K, pv, pi)                  # another query on the handle: the pending first half is void
    with pytest.raises(ValueError):
        eng.scan_partial_bounded(q, B, K, bd, 1, pv, pi)


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
@pytest.mark.parametrize("n,d,m,k,K,B", [(150000, 64, 16, 256, 10, 33), (30000, 24, 6, 16, 5, 9), (5000, 16, 4, 300, 3, 4)])
def test_c_abi_sharded_index_equals_unsharded_and_oracle(oracle, g, devices, n, d, m, k, K, B):
    """gulon_sharded_index_*: a one-device RCCL group (all this box has), one shard or three shards on it."""
    from gulon_amd.sharded import NodeShardedIndex
    cents, idx, pq, enc = _make(g, n, d, m, k, seed=n + K)
    Q = np.random.default_rng(3).standard_normal((B, d)).astype(np.float32)
    sx = NodeShardedIndex(pq, enc, devices)
    info = sx.info()
    assert info["shards"] == len(devices) and info["devices"] == 1 and info["rccl_version"] > 0
    res = sx.batch_query_raw(K, Q)
    full = g.PQIndex(pq, enc).batch_query_raw(K, Q)
    if k <= 256:
        _same(res, full)
    else:   # wide codes: no exact replay on either path; ids agree wherever no tie was flagged
        assert np.array_equal(bits(res[1]), bits(full[1])) and np.array_equal(res[2], full[2])
    _check(res, *oracle.pq_batch_query(idx, d, k, cents, Q, K))
    sx.close()


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_c_abi_sharded_index_every_query_tied(oracle, g, devices):
    """Every query of the batch sits on duplicated rows (first and last shard): far more flagged queries than
    the first replay round holds -- all of them must come back with the literal heap's ids and order."""
    from gulon_amd.sharded import NodeShardedIndex
    n, d, m, k, K, B = 120000, 32, 8, 16, 7, 200
    cents, idx, pq, enc = _make(g, n, d, m, k, seed=12, dup=30000)
    whole = g.PQIndex(pq, enc)
    Q = np.stack([whole.decode(r) for r in range(0, 30000, 150)][:B]).astype(np.float32)
    sx = NodeShardedIndex(pq, enc, devices)
    res = sx.batch_query_raw(K, Q)
    info = sx.info()
    assert info["last_flagged_queries"] == B and info["last_replay_rounds"] == 1 + -(-(B - 16) // 128)
    assert ((res[3] & 3) != 0).all() and ((res[3] & 4) != 0).all()
    _same(res, whole.batch_query_raw(K, Q))
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    assert np.array_equal(res[0], oi) and np.array_equal(bits(res[1]), bits(od))
    sx.close()


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
def test_c_abi_sharded_index_nan_and_inf_queries(oracle, g, devices):
    """A NaN query component makes every distance NaN: the reference returns the first K rows of the index as
    [1, ..., K-1, 0] (TopKHeap.scala:69-79) -- also across shards; +inf distances go through the tie replay."""
    from gulon_amd.sharded import NodeShardedIndex
    n, d, m, k, K, B = 100000, 32, 8, 256, 10, 5
    cents, idx, pq, enc = _make(g, n, d, m, k, seed=31)
    Q = np.random.default_rng(3).standard_normal((B, d)).astype(np.float32)
    Q[1, 7] = np.nan
    Q[3, :] = 1e30
    sx = NodeShardedIndex(pq, enc, devices)
    ri, rd, rc, rf = sx.batch_query_raw(K, Q)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    assert np.array_equal(rc, oc) and np.array_equal(ri, oi)
    assert np.array_equal(np.isnan(rd), np.isnan(od)) and np.array_equal(bits(rd[~np.isnan(od)]), bits(od[~np.isnan(od)]))
    assert rf[1] & 8
    sx.close()


def test_c_abi_sharded_index_argument_errors(g):
    from gulon_amd.sharded import NodeShardedIndex
    cents, idx, pq, enc = _make(g, 1000, 8, 2, 16, seed=1)
    with pytest.raises(ValueError):
        NodeShardedIndex(pq, enc, [99])
    sx = NodeShardedIndex(pq, enc, [0, 0])
    with pytest.raises(NotImplementedError):
        sx.batch_query_raw(8191, np.zeros((1, 8), np.float32))
    oi, od, oc, of = sx.batch_query_raw(3, np.zeros((0, 8), np.float32))
    assert len(oc) == 0
    sx.close()


@pytest.mark.parametrize("devices", [[0], [0, 0, 0]])
@pytest.mark.parametrize("K", [64, 200, 1000])
def test_c_abi_sharded_index_large_k(oracle, g, devices, K):
    """Tests.scala asks an index for up to 1000 neighbours: beyond the 63 a wavefront list holds, every shard peels
    its K+1 best and the lists are merged pairwise -- equal to the unsharded index and (where no distances tie)
    to the oracle."""
    from gulon_amd.sharded import NodeShardedIndex
    n, d, m, k, B = 60000, 32, 8, 256, 5
    cents, idx, pq, enc = _make(g, n, d, m, k, seed=K)
    Q = np.random.default_rng(4).standard_normal((B, d)).astype(np.float32)
    sx = NodeShardedIndex(pq, enc, devices)
    res = sx.batch_query_raw(K, Q)
    full = g.PQIndex(pq, enc).batch_query_raw(K, Q)
    _same(res, full)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    assert np.array_equal(bits(res[1]), bits(od)) and np.array_equal(res[2], oc)
    for q in range(B):
        if res[3][q] == 0:
            assert np.array_equal(res[0][q], oi[q])
    sx.close()


def test_two_indexes_tuned_independently(oracle, g):
    """gulon_index_tuning: the knobs of one handle do not leak into another index of the same process."""
    import ctypes as C
    from gulon_amd import native as N
    n, d, m, k, K, B = 90000, 32, 8, 256, 10, 9
    cents, idx, pq, enc = _make(g, n, d, m, k, seed=77)
    Q = np.random.default_rng(5).standard_normal((B, d)).astype(np.float32)
    a, b = g.PQIndex(pq, enc), g.PQIndex(pq, enc)
    N.check(N.lib().gulon_index_tuning(a._h, b"GULON_SCAN_FILTER", 0))      # a: exact scan only
    exp = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    _check(a.batch_query_raw(K, Q), *exp)
    _check(b.batch_query_raw(K, Q), *exp)
    t, r = C.c_int32(-1), C.c_int32(-1)
    N.check(N.lib().gulon_index_filter_stats(a._h, C.byref(t), C.byref(r)))
    assert t.value == 0                                                    # a's batch took the exact scan ...
    N.check(N.lib().gulon_index_filter_stats(b._h, C.byref(t), C.byref(r)))
    assert t.value > 0                                                     # ... b's the filter
    with pytest.raises(ValueError):
        N.check(N.lib().gulon_index_tuning(a._h, b"NO_SUCH_KNOB", 1))
    a.close(); b.close()
