"""Row shards that share their pruning bounds (gulon_index_scan_bounds_dev -> all-gather ->
gulon_index_scan_partial_bounded_dev, driven by sharded.ShardedIndex): the result must equal the
unsharded index bit for bit, whatever the number of shards and whichever stages the shared bound
retires.  The ranks are threads of this process on the one GPU of the test box, the "all-gather"
a device-to-device copy behind a barrier -- the same ShardedIndex code path as under RCCL."""
import threading

import numpy as np
import pytest

from conftest import bits
from test_gpu_query import _make

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gulon_amd
    assert gulon_amd.native.device_count() >= 1
    return gulon_amd


@pytest.fixture
def tune(g):
    """Knobs for the handles created from now on (the environment) and for the ones already open (gulon_index_tuning)."""
    import os
    defaults = dict(GULON_FILTER_MIN_RB=512, GULON_FILTER_PERIOD=128, GULON_FILTER_STAGE0=0, GULON_FILTER_STAGE1=10,
                    GULON_FILTER_SAMPLE=65536, GULON_FILTER_SHARED_STAGE1=-1)
    before = {k: os.environ.get(k) for k in defaults}
    yield g.tune_live
    g.tune_live(**defaults)
    for k, v in before.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


class ThreadGroup:
    """torch.distributed look-alike for `world` threads of one process (device tensors)."""

    def __init__(self, world):
        import torch
        self.torch, self.world = torch, world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.local = threading.local()

    def get_backend(self):
        return "threads"

    def all_gather_into_tensor(self, out, inp):
        r = self.local.rank
        self.torch.cuda.synchronize()
        self.slots[r] = inp
        self.barrier.wait()
        n = inp.shape[0]
        for s in range(self.world):
            out[s * n:(s + 1) * n].copy_(self.slots[s])
        self.torch.cuda.synchronize()
        self.barrier.wait()


def sharded_query(g, pq, enc, n, world, Q, K, share=True, monkeypatch=None):
    import torch
    from gulon_amd.sharded import HipEngine, ShardedIndex, local_shard, shard_bounds
    grp = ThreadGroup(world)
    out, err = [None] * world, []
    engines = []
    for r in range(world):
        lo, hi = shard_bounds(n, world, r)
        engines.append(HipEngine(pq, local_shard(pq, enc, lo, hi), lo, torch.device("cuda", 0)))

    def run(r):
        try:
            grp.local.rank = r
            sh = ShardedIndex(engines[r], n, r, world, grp)
            sh.share_bounds = share
            out[r] = sh.batch_query(K, Q)
        except Exception as e:          # pragma: no cover
            err.append(e)
            grp.barrier.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not err, err
    for r in range(1, world):           # every rank holds the same answer
        for a, b in zip(out[0], out[r]):
            assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a,
                                  b.view(np.uint32) if b.dtype == np.float32 else b)
    for e in engines:
        e.index.close()
    return out[0]


def _same(a, b):
    assert np.array_equal(a[0], b[0])                       # ids
    assert np.array_equal(bits(a[1]), bits(b[1]))           # distances
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])   # counts, flags


def _trained(g, n, d, m, k, seed=5):
    dm = g.DeviceMatrix.synthetic(n, d, 3, seed, 200)
    pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, 3))
    return dm, pq, pq.encode(dm)


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("stage1", [-1, 0, 1])
def test_shared_bounds_equal_unsharded(g, tune, world, stage1):
    n, d, m, k, B, K = 600000, 64, 16, 256, 70, 10
    tune(GULON_FILTER_SHARED_STAGE1=stage1)
    dm, pq, enc = _trained(g, n, d, m, k)
    Q = dm.get_rows(np.arange(0, n, n // B, dtype=np.int32)[:B])
    full = g.PQIndex(pq, enc).batch_query_raw(K, Q)
    _same(sharded_query(g, pq, enc, n, world, Q, K, share=True), full)


def test_shared_bounds_small_thresholds_every_stage(oracle, g, tune):
    """Oracle-sized shards (thresholds shrunk so that they still take the filter), random codes: the
    bounds separate little, queries give up and fall back -- all of it with the shared bound."""
    n, d, m, k, B, K = 60000, 32, 8, 256, 21, 5
    tune(GULON_FILTER_MIN_RB=4, GULON_FILTER_PERIOD=8, GULON_FILTER_STAGE0=1, GULON_FILTER_STAGE1=2,
         GULON_FILTER_SAMPLE=512)
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=2)
    Q = np.random.default_rng(3).standard_normal((B, d)).astype(np.float32)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    for stage1 in (0, 1):
        tune(GULON_FILTER_SHARED_STAGE1=stage1)
        ri, rd, rc, rf = sharded_query(g, pq, enc, n, 4, Q, K)
        assert np.array_equal(bits(rd), bits(od)) and np.array_equal(rc, oc)
        ok = (rf == 0) | ((rf & 4) != 0)
        assert np.array_equal(ri[ok], oi[ok])


def test_shared_bounds_with_ties_replayed(oracle, g, tune):
    n, d, m, k, B, K = 300000, 64, 16, 256, 12, 10
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=9, dup=5000)
    whole = g.PQIndex(pq, enc)
    # queries = reconstructions of duplicated rows: distance 0 to a row of the first AND of the last shard
    Q = np.stack([whole.decode(r) for r in range(0, 5000, 5000 // B)][:B]).astype(np.float32)
    full = whole.batch_query_raw(K, Q)
    res = sharded_query(g, pq, enc, n, 3, Q, K)
    _same(res, full)
    assert ((res[3] & 3) != 0).any() and (((res[3] & 3) != 0) == ((res[3] & 4) != 0)).all()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_every_query_flagged_across_shards_is_replayed(oracle, g, tune, world):
    """150 queries that all tie across the first and the last shard: the first replay round takes 16, further
    rounds (ShardedIndex.complete) the rest -- ids and order equal the unsharded index's and the oracle's."""
    n, d, m, k, B, K = 200000, 32, 8, 16, 150, 10
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=19, dup=40000)
    whole = g.PQIndex(pq, enc)
    Q = np.stack([whole.decode(r) for r in range(0, 40000, 40000 // B)][:B]).astype(np.float32)
    full = whole.batch_query_raw(K, Q)
    res = sharded_query(g, pq, enc, n, world, Q, K)
    _same(res, full)
    assert ((res[3] & 3) != 0).all() and ((res[3] & 4) != 0).all()
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    assert np.array_equal(res[0], oi) and np.array_equal(bits(res[1]), bits(od))


@pytest.mark.parametrize("K", [64, 300])
def test_large_k_across_shards(g, tune, K):
    """k_nn beyond a wavefront list through ShardedIndex (threads as ranks): peeled partial lists, long merge."""
    n, d, m, k, B = 200000, 32, 8, 256, 7
    dm, pq, enc = _trained(g, n, d, m, k, seed=3)
    Q = dm.get_rows(np.arange(B, dtype=np.int32) * 313)
    _same(sharded_query(g, pq, enc, n, 3, Q, K), g.PQIndex(pq, enc).batch_query_raw(K, Q))


@pytest.mark.parametrize("B,K", [(1, 1), (5, 63), (130, 2)])
def test_shared_bounds_ragged_batches(g, tune, B, K):
    n, d, m, k = 300000, 32, 8, 256
    dm, pq, enc = _trained(g, n, d, m, k, seed=8)
    Q = dm.get_rows(np.arange(B, dtype=np.int32) * 997 % n)
    _same(sharded_query(g, pq, enc, n, 4, Q, K), g.PQIndex(pq, enc).batch_query_raw(K, Q))


def test_shards_too_small_for_the_filter_offer_no_bound(g, tune):
    """20 000-row shards stay on the exact scan: their bounds are +inf, the result is unchanged."""
    import torch
    from gulon_amd import native as N
    from gulon_amd.sharded import HipEngine, local_shard
    n, d, m, k, B, K = 40000, 32, 8, 256, 9, 10
    dm, pq, enc = _trained(g, n, d, m, k, seed=2)
    Q = dm.get_rows(np.arange(B, dtype=np.int32) * 11)
    _same(sharded_query(g, pq, enc, n, 2, Q, K), g.PQIndex(pq, enc).batch_query_raw(K, Q))
    eng = HipEngine(pq, local_shard(pq, enc, 0, 20000), 0, torch.device("cuda", 0))
    bd = eng.alloc((B, K + 1), "f32")
    eng.scan_bounds(eng.to_device(Q), B, K, bd)
    assert torch.isinf(bd).all()


def test_bounds_are_sample_distances_and_bound_the_result(g, tune):
    import torch
    from gulon_amd.sharded import HipEngine
    n, d, m, k, B, K = 400000, 64, 16, 256, 33, 10
    dm, pq, enc = _trained(g, n, d, m, k, seed=6)
    Q = dm.get_rows(np.arange(B, dtype=np.int32) * 1201)
    eng = HipEngine(pq, enc, 0, torch.device("cuda", 0))
    q = eng.to_device(Q)
    bd = eng.alloc((B, K + 1), "f32")
    pv, pi = eng.alloc((B, K + 1), "f32"), eng.alloc((B, K + 1), "i32")
    eng.scan_bounds(q, B, K, bd)
    eng.scan_partial_bounded(q, B, K, bd, 1, pv, pi)
    b, v = bd.cpu().numpy(), pv.cpu().numpy()
    assert (np.diff(b, axis=1) >= 0).all()                  # ascending
    assert (b[:, K] >= v[:, K]).all()                       # the bound is an upper bound of the (K+1)-th distance
    full = g.PQIndex(pq, enc).batch_query_raw(K, Q)
    assert np.array_equal(bits(v[:, :K]), bits(full[1]))


def test_second_half_requires_the_first(g):
    import torch
    from gulon_amd import native as N
    from gulon_amd.sharded import HipEngine
    n, d, m, k, B, K = 100000, 32, 8, 256, 4, 3
    dm, pq, enc = _trained(g, n, d, m, k, seed=1)
    eng = HipEngine(pq, enc, 0, torch.device("cuda", 0))
    q = eng.to_device(dm.get_rows(np.arange(B, dtype=np.int32)))
    bd = eng.alloc((B, K + 1), "f32")
    pv, pi = eng.alloc((B, K + 1), "f32"), eng.alloc((B, K + 1), "i32")
    with pytest.raises(ValueError):
        eng.scan_partial_bounded(q, B, K, bd, 1, pv, pi)
    eng.scan_bounds(q, B, K, bd)
    with pytest.raises(ValueError):                       # different k_nn than the first half
        eng.scan_partial_bounded(q, B, K - 1, bd, 1, pv, pi)
