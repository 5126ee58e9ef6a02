"""World-size-2 test of the row-sharded query orchestration on CPU (gloo backend):
shard bounds, per-shard row bases, all-gather layout and merge inputs.  The compute engine
is a test double built on the CPU oracle (the shipped engine is HipEngine, exercised by
tests/test_gpu_query.py::test_sharded_partials_merge_equals_full on one GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

INT_MAX = 2 ** 31 - 1


FLAGGED = 4     # flagged queries the test double replays in the first round
MORE = 6        # ... and in every further round (ShardedIndex.complete)


class OracleEngine:
    def __init__(self, oracle, idx_local, d, k, cents, row_base, cap=None):
        self.o, self.idx, self.d, self.k, self.cents, self.row_base = oracle, idx_local, d, k, cents, row_base
        self.nloc = idx_local.shape[1]
        self.cap = cap or self.nloc          # rows of the largest shard: every rank's buffer has the same size
        self.replay_first, self.replay_more = FLAGGED, MORE

    def replay_words(self, F):
        # candidate buffer: [nflag here][rows here][flagged in the batch][F qids][F x cap distance bits][F x cap row ids]
        return 3 + F + 2 * F * self.cap

    def alloc(self, shape, dtype):
        return torch.empty(shape, dtype={"f32": torch.float32, "i32": torch.int32}[dtype])

    def to_device(self, a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def scan_partial(self, q, b, k, pv, pi):
        oi, od, oc = self.o.pq_batch_query(self.idx, self.d, self.k, self.cents, q.numpy(), k + 1)
        v = np.full((b, k + 1), np.inf, np.float32)
        i = np.full((b, k + 1), INT_MAX, np.int32)
        for r in range(b):
            v[r, :oc[r]] = od[r, :oc[r]]
            i[r, :oc[r]] = oi[r, :oc[r]] + self.row_base
        pv.copy_(torch.from_numpy(v))
        pi.copy_(torch.from_numpy(i))

    def views(self, pk, b):
        return pk[:b].view(torch.float32), pk[b:]

    def merge(self, packed, lists, b, k, oi, od, oc, of):
        w = packed.numpy().reshape(lists, 2, b, k + 1)
        av = np.ascontiguousarray(w[:, 0]).view(np.float32)
        ai = np.ascontiguousarray(w[:, 1])
        for r in range(b):
            cand = sorted((float(av[l, r, e]), int(ai[l, r, e])) for l in range(lists) for e in range(k + 1)
                          if ai[l, r, e] != INT_MAX)
            top = cand[:k]
            oc[r] = len(top)
            f = 0
            if len(cand) > k and cand[k][0] == cand[k - 1][0]:
                f |= 1                                            # GULON_FLAG_BOUNDARY_TIE
            if any(top[e][0] == top[e + 1][0] for e in range(len(top) - 1)):
                f |= 2                                            # GULON_FLAG_INTERIOR_TIE
            of[r] = f
            for e, (v, i) in enumerate(top):
                od[r, e] = v
                oi[r, e] = i

    def replay_collect(self, q, b, k, of, pack, skip, F):
        """Every local row is a candidate (a superset of the inserting rows), with its exact distance."""
        every = [r for r in range(b) if int(of[r]) & 3]
        flagged = every[skip:skip + F]
        w = np.zeros(self.replay_words(F), np.int32)
        w[0], w[1], w[2] = len(flagged), self.nloc, len(every)
        w[3:3 + F] = -1
        T = self.o.prepare_query(self.cents, self.d, self.idx.shape[0], self.k, q.numpy())
        m, n = self.idx.shape
        for f, r in enumerate(flagged):
            w[3 + f] = r
            acc = np.zeros(n, np.float32)
            for j in range(m):                                    # the reference's order: j ascending, fp32
                acc = (acc + T[r, j, self.idx[j]]).astype(np.float32)
            o = 3 + F + f * self.cap
            w[o:o + n] = acc.view(np.int32)
            o2 = 3 + F + F * self.cap + f * self.cap
            w[o2:o2 + n] = np.arange(n, dtype=np.int32) + self.row_base
        pack.copy_(torch.from_numpy(w))

    def replay_total(self, pack):
        return int(pack[2])

    def replay_apply(self, packs, lists, b, k, oi, od, oc, of, F):
        """Literal TopKHeap over the union of the shards' candidates, in row order."""
        w = packs.numpy()
        per = len(w) // lists
        nflag = int(w[0])
        for f in range(nflag):
            r = int(w[3 + f])
            rows, dists = [], []
            for l in range(lists):
                wl = w[l * per:(l + 1) * per]
                n = int(wl[1])
                o = 3 + F + f * self.cap
                o2 = 3 + F + F * self.cap + f * self.cap
                rows.append(wl[o2:o2 + n])
                dists.append(wl[o:o + n].view(np.float32))
            rows, dists = np.concatenate(rows), np.concatenate(dists)
            order = np.argsort(rows, kind="stable")
            h = self.o.TopKHeap(k)
            for e in order:
                h.update(int(rows[e]), dists[e])
            ks, vs = h.drain()
            oc[r] = len(ks)
            for e in range(len(ks)):
                oi[r, e] = int(ks[e])
                od[r, e] = float(vs[e])
            of[r] = int(of[r]) | 4                                # GULON_FLAG_EXACT_REPLAY


class BoundedOracleEngine(OracleEngine):
    """The two halves of the scan with shared pruning bounds (gulon_index_scan_bounds_dev /
    gulon_index_scan_partial_bounded_dev): sample distances out, then only rows within the bound of the
    union of all shards' samples -- the partial list may come back shorter than K+1."""

    def _all(self, q):
        T = self.o.prepare_query(self.cents, self.d, self.idx.shape[0], self.k, q.numpy())
        m, n = self.idx.shape
        acc = np.zeros((q.shape[0], n), np.float32)
        for j in range(m):                                        # the reference's order: j ascending, fp32
            acc = (acc + T[:, j, self.idx[j]]).astype(np.float32)
        return acc

    def scan_bounds(self, q, b, k, bd):
        d = np.sort(self._all(q)[:, ::7], axis=1)[:, :k + 1]      # every 7th row is the sample
        v = np.full((b, k + 1), np.inf, np.float32)
        v[:, :d.shape[1]] = d
        bd.copy_(torch.from_numpy(v))
        self.pending = (b, k)

    def scan_partial_bounded(self, q, b, k, abd, lists, pv, pi):
        assert self.pending == (b, k)
        tau = np.sort(abd.numpy().reshape(lists, b, k + 1).transpose(1, 0, 2).reshape(b, -1), axis=1)[:, k]
        acc = self._all(q)
        v = np.full((b, k + 1), np.inf, np.float32)
        i = np.full((b, k + 1), INT_MAX, np.int32)
        for r in range(b):
            keep = np.nonzero(acc[r] <= tau[r])[0]
            order = keep[np.lexsort((keep, acc[r, keep]))][:k + 1]
            v[r, :len(order)] = acc[r, order]
            i[r, :len(order)] = order + self.row_base
        self.shortest = int(min((i[r] != INT_MAX).sum() for r in range(b)))
        pv.copy_(torch.from_numpy(v))
        pi.copy_(torch.from_numpy(i))


def _worker(rank, world, port, n, d, m, k, B, K, out, dup=0, bounded=False, flagged_min=0, dup_queries=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle
    from gulon_amd.sharded import ShardedIndex, shard_bounds
    rng = np.random.default_rng(0)
    cents = rng.standard_normal(k * d).astype(np.float32)
    idx = rng.integers(0, k, (m, n)).astype(np.int32)
    if dup:
        idx[:, -dup:] = idx[:, :dup]          # identical codes in the first and the last shard: distance ties
    Q = rng.standard_normal((B, d)).astype(np.float32)
    if dup_queries:                           # every query sits exactly on a duplicated row: all of them tie
        fr, un = oracle.subvectors(d, m)
        for r in range(B):
            for j in range(m):
                Q[r, fr[j]:un[j]] = cents[k * fr[j] + idx[j, r] * (un[j] - fr[j]): k * fr[j] + (idx[j, r] + 1) * (un[j] - fr[j])]
    lo, hi = shard_bounds(n, world, rank)
    eng = (BoundedOracleEngine if bounded else OracleEngine)(oracle, np.ascontiguousarray(idx[:, lo:hi]), d, k, cents, lo,
                                                             cap=-(-n // world))
    sh = ShardedIndex(eng, n, rank, world, dist)
    assert sh.share_bounds == bounded
    oi, od, oc, of = sh.batch_query(K, Q)
    ei, ed, ec = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    ok = bool(np.array_equal(od.view(np.uint32), ed.view(np.uint32)) and np.array_equal(oc, ec))
    for r in range(B):      # ids: the reference's wherever there is no tie or the tie was replayed
        if of[r] == 0 or (of[r] & 4):
            ok = ok and bool(np.array_equal(oi[r], ei[r]))
    if dup:     # ties exist, and EVERY flagged query was replayed (more of them than one round holds: complete())
        ok = ok and bool(((of & 3) != 0).any()) and bool((((of & 3) != 0) == ((of & 4) != 0)).all())
        if flagged_min:
            ok = ok and int(((of & 3) != 0).sum()) >= flagged_min and sh.last_flagged >= flagged_min
    t = torch.tensor([1 if ok else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        out.put(int(t.item()))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n", [(2, 5001), (3, 1000)])
def test_sharded_query_equals_unsharded(world, n):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, 12, 4, 16, 5, 7, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=5) == 1


@pytest.mark.parametrize("world,n,dup", [(2, 3001, 0), (3, 2000, 0), (2, 900, 300)])
def test_sharded_query_with_shared_bounds(world, n, dup):
    """Three collectives per batch: bounds, partial lists, replay candidates (sharded.ShardedIndex with an
    engine that splits its scan) -- results equal the unsharded oracle, also under ties."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, 8, 2, 16, 4, 5, out, dup, True)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=5) == 1


def test_sharded_tie_replay_equals_reference_heap():
    """Duplicate rows across shards => tie flags => candidates of all shards gathered and replayed
    through the literal TopKHeap: ids and order equal the unsharded reference semantics."""
    world, n = 2, 600
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, 8, 2, 4, 3, 5, out, 250)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=5) == 1


@pytest.mark.parametrize("world,bounded", [(2, False), (3, True)])
def test_sharded_tie_replay_more_flagged_than_one_round(world, bounded):
    """17 queries that all tie across shards, 4 per first round and 6 per further round in the test double:
    ShardedIndex.complete() must keep exchanging candidates until every flagged query has been replayed."""
    n, B = 600, 17
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, 8, 2, 4, B, 5, out, 250, bounded, B, True))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=5) == 1


def test_shard_bounds_tile_the_rows():
    from gulon_amd.sharded import shard_bounds
    for n in (0, 1, 7, 10_000_000, 9_999_999):
        for w in (1, 2, 3, 4, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
