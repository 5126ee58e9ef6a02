"""GPU parity tests for the quantized lower-bound filter in front of the ADC scan (filter.hip).

The filter only changes how much work is done, never a result: everything here must equal the
CPU oracle bit for bit, exactly as the unfiltered scan does.  The filter is normally reserved for
large row ranges; the tuning knobs shrink its thresholds so that small, oracle-sized cases run
through every stage (sample bounds, three filter stages, survivor re-evaluation, fallback)."""
import numpy as np
import pytest

from conftest import bits
from test_gpu_query import _check, _make

pytestmark = pytest.mark.gpu

DEFAULTS = {"GULON_SCAN_FILTER": 1, "GULON_FILTER_ORDER": 1, "GULON_FILTER_MIN_RB": 512, "GULON_FILTER_PERIOD": 128,
            "GULON_FILTER_STAGE0": 0, "GULON_FILTER_STAGE1": 10, "GULON_FILTER_CAP": 32768,
            "GULON_FILTER_NADD": 0, "GULON_FILTER_SAMPLE": 65536}


@pytest.fixture(scope="module")
def g():
    import gulon_amd
    assert gulon_amd.native.device_count() >= 1
    return gulon_amd


@pytest.fixture
def tune(g):
    """Knobs for the handles created from now on (the environment) and for the ones already open (gulon_index_tuning)."""
    import os
    before = {k: os.environ.get(k) for k in DEFAULTS}
    g.tune_live(GULON_FILTER_MIN_RB=4, GULON_FILTER_PERIOD=8, GULON_FILTER_STAGE0=1, GULON_FILTER_STAGE1=2,
                GULON_FILTER_SAMPLE=512)
    yield g.tune_live
    g.tune_live(**DEFAULTS)
    for k, v in before.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def test_tuning_rejects_unknown_key(oracle, g):
    from gulon_amd import native as N
    cents, idx, pq, enc = _make(oracle, g, 300, 8, 2, 4, seed=1)
    ix = g.PQIndex(pq, enc)
    assert N.lib().gulon_index_tuning(ix._h, b"GULON_NO_SUCH_KNOB", 1) != 0
    assert N.lib().gulon_index_tuning(ix._h, b"GULON_FILTER_CAP", 128) == 0
    ix.close()


@pytest.mark.parametrize("nadd", [2, 4])
@pytest.mark.parametrize("n,d,m,k,B,K,frm,until", [
    (100000, 128, 16, 256, 37, 10, 0, None),        # BASELINE shape: 2 groups of 16 queries per workgroup
    (60000, 128, 16, 256, 100, 10, 0, None),        # several filter tiles, ragged last one
    (30000, 100, 25, 256, 9, 10, 0, None),          # ragged m: 4-byte code words, one group per workgroup
    (40000, 64, 32, 256, 20, 5, 0, None),           # m = 32: two 16-byte code words per row
    (40000, 32, 8, 256, 33, 1, 0, None),            # K = 1
    (20000, 12, 4, 5, 3, 20, 0, None),              # width-4 codes, k = 5
    (20000, 8, 8, 1, 2, 7, 0, None),                # k = 1: every row has the same distance
    (20000, 16, 4, 3, 4, 63, 0, None),              # width-2 codes, max K
    (50000, 128, 16, 256, 8, 10, 12345, 40001),     # from/until sub-range (partial first and last block)
    (20000, 72, 36, 256, 5, 10, 0, None),           # widest 16-queries-per-entry table (m_pad = 36)
    (30000, 128, 64, 256, 21, 10, 0, None),         # m = 64: 8 queries per entry (ds_read_b64), float2 exact tables
    (20000, 96, 48, 37, 9, 10, 77, 15000),          # m = 48, k = 37, sub-range
    (12000, 200, 100, 256, 7, 10, 0, None),         # m = 100: 4 queries per entry (ds_read_b32), scalar exact tables
    (12000, 80, 40, 256, 35, 3, 0, None),           # m = 40: 8 per entry, several tiles
])
def test_filtered_query_bit_exact(oracle, g, tune, n, d, m, k, B, K, frm, until, nadd):
    tune(GULON_FILTER_NADD=nadd)
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=n + d)
    Q = np.random.default_rng(7).standard_normal((B, d)).astype(np.float32)
    ix = g.PQIndex(pq, enc)
    res = ix.batch_query(K, Q, frm, until)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K, frm, n if until is None else until)
    _check(oracle, res, oi, od, oc)
    ix.close()


def test_filter_equals_unfiltered_scan(oracle, g, tune):
    """Same index, same queries: filter on vs off give identical ids, distances and flags."""
    n, d, m, k, B, K = 200000, 128, 16, 256, 64, 10
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=5)
    Q = np.random.default_rng(11).standard_normal((B, d)).astype(np.float32)
    ix = g.PQIndex(pq, enc)
    a = ix.batch_query(K, Q)
    tune(GULON_SCAN_FILTER=0)
    b = ix.batch_query(K, Q)
    for x, y in zip(a, b):
        assert x.rows.tolist() == y.rows.tolist()
        assert np.array_equal(bits(x.distances), bits(y.distances))
        assert x.flags == y.flags
    ix.close()


def test_fallback_launch_width_adapts_over_repeated_batches(oracle, g, tune):
    """Random codes and a 512-row sample: most query tiles give up and are redone by the exact scan.  The
    fallback launch starts narrow (8 looping workgroups), turns wide once a launch reported work and stays
    so -- every batch of the series equals the unfiltered scan."""
    n, d, m, k, B, K = 200000, 128, 16, 256, 64, 10
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=5)
    Q = np.random.default_rng(11).standard_normal((B, d)).astype(np.float32)
    ix = g.PQIndex(pq, enc)
    tune(GULON_SCAN_FILTER=0)
    ref = ix.batch_query(K, Q)
    tune(GULON_SCAN_FILTER=1, GULON_FILTER_CAP=64)
    for _ in range(5):
        got = ix.batch_query(K, Q)
        for x, y in zip(got, ref):
            assert x.rows.tolist() == y.rows.tolist()
            assert np.array_equal(bits(x.distances), bits(y.distances))
            assert x.flags == y.flags
    ix.close()


def test_queries_that_are_dataset_rows(oracle, g, tune):
    """A query equal to a centroid combination has distance == sum of the table minima (budget 0)."""
    n, d, m, k, B, K = 50000, 64, 16, 256, 12, 10
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=21)
    from gulon_amd.vectors import subvector_bounds
    fr, un = subvector_bounds(d, m)
    Q = np.zeros((B, d), np.float32)
    for q in range(B):           # decode row q: its ADC distance to itself is exactly 0
        for j in range(m):
            s = un[j] - fr[j]
            c = idx[j, q]
            Q[q, fr[j]:un[j]] = cents[k * fr[j] + c * s: k * fr[j] + (c + 1) * s]
    ix = g.PQIndex(pq, enc)
    res = ix.batch_query(K, Q)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    _check(oracle, res, oi, od, oc)
    assert all(r.distances[0] == 0.0 for r in res)
    ix.close()


def test_ties_are_flagged_and_replayed(oracle, g, tune):
    n, d, m, k, B, K = 60000, 128, 16, 256, 10, 10
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=3, dup=3000)
    Q = np.random.default_rng(2).standard_normal((B, d)).astype(np.float32)
    ix = g.PQIndex(pq, enc)
    res = ix.batch_query(K, Q)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    _check(oracle, res, oi, od, oc)
    ix.close()


def test_queue_overflow_falls_back_to_exact_scan(oracle, g, tune):
    """64-entry survivor sub-queues (1024 per query) overflow for every query; the device-side
    fallback redoes them with the exact scan."""
    tune(GULON_FILTER_CAP=64)
    n, d, m, k, B, K = 80000, 16, 4, 2, 21, 10           # 16 distinct codes: ~5000 rows per distance value
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=9)
    Q = np.random.default_rng(4).standard_normal((B, d)).astype(np.float32)
    ix = g.PQIndex(pq, enc)
    res = ix.batch_query(K, Q)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    _check(oracle, res, oi, od, oc)
    ix.close()


def _check_nonfinite(res, oi, od, oc):
    """ids and order equal the oracle's for every query; distances bit for bit, NaNs as NaNs."""
    for q, r in enumerate(res):
        assert len(r) == oc[q]
        assert r.rows.tolist() == oi[q, :oc[q]].tolist(), q
        e = od[q, :oc[q]]
        assert np.array_equal(np.isnan(r.distances), np.isnan(e))
        ok = ~np.isnan(e)
        assert np.array_equal(bits(r.distances[ok]), bits(e[ok]))


@pytest.mark.parametrize("n,frm,until", [(30000, 0, None), (90000, 0, None), (90000, 1234, 80001), (300, 10, 15)])
def test_nan_and_huge_queries(oracle, g, tune, n, frm, until):
    """TopKHeap.update inserts a NaN while the heap is not full and nothing afterwards (TopKHeap.scala:69-79): a
    query with a NaN component gets the first K rows of the range, drained as [1, ..., K-1, 0]; a query whose
    distances all overflow to +inf likewise.  Compared with the oracle, query by query."""
    d, m, k, B, K = 64, 16, 256, 6, 10
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=13)
    Q = np.random.default_rng(5).standard_normal((B, d)).astype(np.float32)
    Q[1, 3] = np.nan            # every distance NaN
    Q[4, :] = 1e30              # every distance overflows to +inf
    Q[5, 60] = np.inf           # inf - c = inf, squared inf: every distance +inf
    ix = g.PQIndex(pq, enc)
    until = n if until is None else until
    res = ix.batch_query(K, Q, frm, until)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K, frm, until)
    _check_nonfinite(res, oi, od, oc)
    assert res[1].flags & 8 and res[4].flags & 8 and not (res[0].flags & 8)
    assert np.isnan(res[1].distances).all() and res[1].rows.tolist() == [frm + (e + 1) % len(res[1]) for e in range(len(res[1]))]
    ix.close()


@pytest.mark.parametrize("n", [5000, 70000])
def test_nan_centroids_mixed_distances(oracle, g, tune, n):
    """NaN / inf codebook entries: some rows have NaN distances, the others ordinary ones -- the heap is no longer
    a heap once a NaN sits in it, and the result is whatever the reference's update sequence leaves."""
    d, m, k, B, K = 16, 4, 16, 7, 6
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=21)
    cents = cents.copy()
    cents[5] = np.nan                       # quantizer 0, centroid 1
    cents[k * 8 + 3 * 4 + 2] = np.inf       # quantizer 2 (from = 8, s = 4), centroid 3
    pq = g.ProductQuantizer.from_flat(k, d, m, cents)
    Q = np.random.default_rng(8).standard_normal((B, d)).astype(np.float32)
    ix = g.PQIndex(pq, enc)
    res = ix.batch_query(K, Q)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    _check_nonfinite(res, oi, od, oc)
    assert all(r.flags & 8 for r in res)
    ix.close()


def test_sharded_partial_lists_through_the_filter(oracle, g, tune):
    """Row shards scanned with the filter, merged: equals the single-index result."""
    import ctypes as C
    from gulon_amd import native as N
    from gulon_amd.sharded import local_shard
    n, d, m, k, B, K = 90000, 128, 16, 256, 19, 10
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=17)
    Q = np.random.default_rng(6).standard_normal((B, d)).astype(np.float32)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    L = N.lib()
    bounds = [0, 30000, 61111, n]
    pd, pi = [], []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        ix = g.PQIndex(pq, local_shard(pq, enc, lo, hi), row_base=lo)
        dq, dv, di = C.c_void_p(), C.c_void_p(), C.c_void_p()
        N.check(L.gulon_dev_malloc(C.byref(dq), Q.nbytes))
        N.check(L.gulon_dev_malloc(C.byref(dv), B * (K + 1) * 4))
        N.check(L.gulon_dev_malloc(C.byref(di), B * (K + 1) * 4))
        N.check(L.gulon_memcpy_h2d(dq, Q.ctypes.data_as(C.c_void_p), Q.nbytes))
        N.check(L.gulon_index_scan_partial_dev(ix._h, dq, B, K, 0, hi - lo, dv, di, None))
        N.check(L.gulon_device_synchronize())
        v = np.zeros((B, K + 1), np.float32)
        i = np.zeros((B, K + 1), np.int32)
        N.check(L.gulon_memcpy_d2h(v.ctypes.data_as(C.c_void_p), dv, v.nbytes))
        N.check(L.gulon_memcpy_d2h(i.ctypes.data_as(C.c_void_p), di, i.nbytes))
        for p_ in (dq, dv, di):
            N.check(L.gulon_dev_free(p_))
        pd.append(v)
        pi.append(i)
        ix.close()
    from gulon_amd.topk import merge_partials
    out_i, out_d, out_c, out_f = merge_partials(np.stack(pd), np.stack(pi), K)
    assert np.array_equal(out_i, oi) and np.array_equal(bits(out_d), bits(od))


def test_randomised_shapes_against_the_oracle(oracle, g, tune):
    """Differential test over seeded random shapes: ragged m, small k, sub-ranges, duplicate rows,
    queries on and off the data -- filtered scan == oracle, bit for bit."""
    rng = np.random.default_rng(20260904)
    for case in range(24):
        d = int(rng.integers(4, 97))
        m = int(rng.integers(1, min(d, 40) + 1))
        k = int(rng.choice([2, 3, 16, 17, 100, 256]))
        n = int(rng.integers(600, 40000))
        B = int(rng.integers(1, 40))
        K = int(rng.choice([1, 2, 10, 31, 63]))
        dup = int(rng.integers(0, n // 4)) if case % 3 == 0 else 0
        frm = int(rng.integers(0, n // 3)) if case % 4 == 1 else 0
        until = int(rng.integers(frm + (n - frm) // 2, n + 1)) if case % 4 == 1 else n
        tune(GULON_FILTER_NADD=int(rng.choice([0, 2, 4])), GULON_FILTER_PERIOD=int(rng.choice([4, 8, 32])),
             GULON_FILTER_STAGE1=int(rng.choice([1, 2, 3])))
        cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=1000 + case, dup=dup)
        Q = rng.standard_normal((B, d)).astype(np.float32)
        ix = g.PQIndex(pq, enc)
        res = ix.batch_query(K, Q, frm, until)
        oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K, frm, until)
        try:
            _check(oracle, res, oi, od, oc)
        except AssertionError as e:
            raise AssertionError(f"case {case}: n={n} d={d} m={m} k={k} B={B} K={K} dup={dup} range=[{frm},{until})") from e
        ix.close()


def _two_streams_worker(out):
    """Runs in a fresh process with torch initialised first (the order bench.py uses: torch's HIP
    runtime does not come up after libgulon_hip.so has initialised the device)."""
    import ctypes as C
    import os
    import sys
    import torch
    torch.cuda.init()
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import gulon_amd as g
    from gulon_amd import native as N
    from oracle import oracle
    from test_gpu_query import _make
    n, d, m, k, B, K = 300000, 64, 16, 256, 96, 10           # default thresholds: the filter is active
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=77)
    rng = np.random.default_rng(9)
    Qs = [rng.standard_normal((B, d)).astype(np.float32) for _ in range(2)]
    exp = [oracle.pq_batch_query(idx, d, k, cents, q, K) for q in Qs]
    ixs = [g.PQIndex(pq, enc) for _ in range(2)]
    dq = [torch.from_numpy(q).cuda() for q in Qs]
    streams = [torch.cuda.Stream() for _ in range(2)]
    outs = [[torch.empty((B, K), dtype=torch.int32, device="cuda"), torch.empty((B, K), dtype=torch.float32, device="cuda"),
             torch.empty(B, dtype=torch.int32, device="cuda"), torch.empty(B, dtype=torch.int32, device="cuda")]
            for _ in range(2)]
    torch.cuda.synchronize()
    L = N.lib()
    for rep in range(6):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                oi, od, oc, of = outs[i]
                N.check(L.gulon_index_batch_query_dev(ixs[i]._h, dq[i].data_ptr(), B, K, 0, n, oi.data_ptr(), od.data_ptr(),
                                                      oc.data_ptr(), of.data_ptr(),
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    ok = True
    for i in range(2):
        oi, od, oc, of = (t.cpu().numpy() for t in outs[i])
        ei, ed, ec = exp[i]
        ok = ok and np.array_equal(od.view(np.uint32), ed.view(np.uint32)) and np.array_equal(oc, ec)
        for q in range(B):
            if of[q] == 0 or (of[q] & 4):
                ok = ok and oi[q].tolist() == ei[q].tolist()
    out.put(1 if ok else 0)


def test_two_batches_in_flight_on_two_streams():
    """Two indexes over the same rows, two streams, calls interleaved without synchronising in
    between (what bench.py does): the turn-taking of the main filter stage must not mix results."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_two_streams_worker, args=(out,))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    assert out.get(timeout=5) == 1


# ---- the conflict-ordered copy of the codes (conflict_order.hip) ---------------------------------------------------
LDS_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
LDS_GROUPS += [[lane + 32 for lane in grp] for grp in LDS_GROUPS]


def _gather_cycles(block, nq=13):
    """The measured cost model of the filter's table gathers (scripts/micro/lds_pattern_fit.py): per quantizer, the sum
    over the four lane groups of the most distinct codes in one bank column (code mod 16)."""
    total = 0
    for q in range(nq):
        for grp in LDS_GROUPS:
            codes = set(int(c) for c in block[grp, q])
            total += np.bincount([c % 16 for c in codes], minlength=16).max()
    return total


def test_conflict_order_is_a_permutation_and_lowers_the_model_cost(g):
    """The ordering deals the rows of a WINDOW of four blocks (256 rows; a last window of fewer blocks: of the blocks it
    has) to the lanes of those blocks: place_out[block][lane] = the row's place in its window."""
    from gulon_amd import native as N
    nblk = 602                                     # 150 windows + one of two blocks
    rng = np.random.default_rng(3)
    codes = rng.integers(0, 256, (nblk, 64, 16), dtype=np.uint8)
    codes[20:24] = 7                               # a window whose rows are all alike
    codes[24:28, :, :] = codes[24, :1, :]
    codes[28:32, :, 3] = np.arange(64) * 16 % 256  # one quantizer whose codes all share a bank column
    out, place = np.empty_like(codes), np.empty((nblk, 64), np.uint8)
    for rounds in (0, 1, 2):
        assert 0 == N.hooks_lib().gulon_selftest_conflict_order(codes.ctypes.data, nblk, rounds, out.ctypes.data, place.ctypes.data)
        for w0 in range(0, nblk, 4):
            w1 = min(nblk, w0 + 4)
            rows = codes[w0:w1].reshape(-1, 16)
            pl = place[w0:w1].reshape(-1).astype(np.int64)
            assert np.array_equal(np.sort(pl), np.arange(len(rows)))                 # a permutation of the window's rows
            assert np.array_equal(out[w0:w1].reshape(-1, 16), rows[pl])
        if rounds == 0:
            assert np.array_equal(place[:600].reshape(150, 256), np.tile(np.arange(256, dtype=np.uint8), (150, 1)))
            continue
        before = sum(_gather_cycles(codes[b]) for b in range(100, 300))
        after = sum(_gather_cycles(out[b]) for b in range(100, 300))
        assert after < 0.76 * before, (before, after)       # simulated: 12.3 -> 8.7 cycles per gather over a 256-row window
        for w0 in (20, 24, 28):
            assert sum(_gather_cycles(out[b]) for b in range(w0, w0 + 4)) <= sum(_gather_cycles(codes[b]) for b in range(w0, w0 + 4))


@pytest.mark.parametrize("n,frm,until", [(70001, 0, None), (70001, 12345, 60001), (40000, 63, 39937), (33000, 1000, 1100),
                                         (70100, 257, 69900), (65600, 130, 65599), (40000, 3 * 64, 39000)])
def test_conflict_ordered_copy_equals_plain_codes(oracle, g, tune, n, frm, until):
    """The filter reads its own re-dealt copy of the codes (rows dealt over windows of four blocks) and maps surviving
    lanes back to rows; ranges that cut blocks and windows, a ragged last block, a last window of fewer than four
    blocks: the same answers as with the plain copy (per-handle switch) and as the oracle."""
    from gulon_amd import native as N
    d, m, k, B, K = 128, 16, 256, 40, 10
    cents, idx, pq, enc = _make(oracle, g, n, d, m, k, seed=n)
    Q = np.random.default_rng(5).standard_normal((B, d)).astype(np.float32)
    ix = g.PQIndex(pq, enc)
    ordered = ix.batch_query(K, Q, frm, until)
    N.check(N.lib().gulon_index_tuning(ix._h, b"GULON_FILTER_ORDER", 0))
    plain = ix.batch_query(K, Q, frm, until)
    for x, y in zip(ordered, plain):
        assert x.rows.tolist() == y.rows.tolist()
        assert np.array_equal(bits(x.distances), bits(y.distances))
        assert x.flags == y.flags
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K, frm, n if until is None else until)
    _check(oracle, ordered, oi, od, oc)
    ix.close()
