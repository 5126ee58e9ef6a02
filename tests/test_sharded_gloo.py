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


class OracleEngine:
    def __init__(self, oracle, idx_local, d, k, cents, row_base):
        self.o, self.idx, self.d, self.k, self.cents, self.row_base = oracle, idx_local, d, k, cents, row_base

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
            of[r] = 0
            for e, (v, i) in enumerate(top):
                od[r, e] = v
                oi[r, e] = i


def _worker(rank, world, port, n, d, m, k, B, K, out):
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
    Q = rng.standard_normal((B, d)).astype(np.float32)
    lo, hi = shard_bounds(n, world, rank)
    eng = OracleEngine(oracle, np.ascontiguousarray(idx[:, lo:hi]), d, k, cents, lo)
    sh = ShardedIndex(eng, n, rank, world, dist)
    oi, od, oc, of = sh.batch_query(K, Q)
    ei, ed, ec = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    ok = bool(np.array_equal(oi, ei) and np.array_equal(od.view(np.uint32), ed.view(np.uint32)) and
              np.array_equal(oc, ec))
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


def test_shard_bounds_tile_the_rows():
    from gulon_amd.sharded import shard_bounds
    for n in (0, 1, 7, 10_000_000, 9_999_999):
        for w in (1, 2, 3, 4, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
