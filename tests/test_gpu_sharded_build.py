"""Two ranks (gloo rendezvous, both on the one GPU of the test box) run the quantizer-partitioned
build and the row-sharded query with the real HIP engine; results must equal the single-process
build and query bit for bit.  (RCCL refuses two ranks on one device, so the collectives are
staged through the host here; the device path is what bench.py uses on a multi-GPU node.)"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out):
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gulon_amd as g
    from gulon_amd.sharded import HipEngine, ShardedIndex, build_sharded
    n, d, m, k, iters, B, K = 30000, 48, 6, 256, 3, 9, 10
    dm = g.DeviceMatrix.synthetic(n, d, 3, 77, 30)
    pq, shard, lo, hi = build_sharded(dm, k, m, iters, rank, world, dist, None)
    ref_pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, iters))
    ref_enc = ref_pq.encode(dm)
    ok = np.array_equal(pq.flat_centroids().view(np.uint32), ref_pq.flat_centroids().view(np.uint32))
    ok = ok and all(np.array_equal(shard.encodings[j], ref_enc.encodings[j][lo:hi]) for j in range(m))
    Q = dm.get_rows(np.arange(0, n, n // B, dtype=np.int32)[:B])
    eng = HipEngine(pq, shard, lo, torch.device("cuda", 0))
    oi, od, oc, of = ShardedIndex(eng, n, rank, world, dist).batch_query(K, Q)
    full = g.PQIndex(ref_pq, ref_enc).batch_query_raw(K, Q)
    ok = ok and np.array_equal(oi, full[0]) and np.array_equal(od.view(np.uint32), full[1].view(np.uint32))
    t = torch.tensor([1 if ok else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        out.put(int(t.item()))
    dist.destroy_process_group()


def _tie_worker(rank, world, port, out):
    """Row shards whose rows repeat across the shard boundary: tie-flagged queries must come back with
    the reference's TopKHeap ids and order (GULON_FLAG_EXACT_REPLAY), as from the unsharded index."""
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gulon_amd as g
    from gulon_amd.sharded import HipEngine, ShardedIndex, local_shard, shard_bounds
    from oracle import oracle
    n, d, m, k, B, K = 40000, 64, 16, 256, 12, 10
    rng = np.random.default_rng(3)
    cents = rng.standard_normal(k * d).astype(np.float32)
    idx = rng.integers(0, k, (m, n)).astype(np.int32)
    idx[:, -3000:] = idx[:, :3000]                       # the last rows duplicate the first ones
    pq = g.ProductQuantizer.from_flat(k, d, m, cents)
    coder = pq.coder_factory(n)
    enc = g.EncodedMatrix(coder, [coder.build_code(idx[j]) for j in range(m)])
    Q = rng.standard_normal((B, d)).astype(np.float32)
    lo, hi = shard_bounds(n, world, rank)
    eng = HipEngine(pq, local_shard(pq, enc, lo, hi), lo, torch.device("cuda", 0))
    oi, od, oc, of = ShardedIndex(eng, n, rank, world, dist).batch_query(K, Q)
    ei, ed, ec = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    ok = np.array_equal(od.view(np.uint32), ed.view(np.uint32)) and np.array_equal(oc, ec)
    ok = ok and bool(((of & 3) != 0).any())              # there are ties ...
    ok = ok and bool((((of & 3) != 0) == ((of & 4) != 0)).all())   # ... and every one of them was replayed
    ok = ok and np.array_equal(oi, ei)                   # ids and order: the reference's
    full = g.PQIndex(pq, enc).batch_query_raw(K, Q)
    ok = ok and np.array_equal(oi, full[0]) and np.array_equal(of, full[3])
    t = torch.tensor([1 if ok else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        out.put(int(t.item()))
    dist.destroy_process_group()


def _run(target, world=2):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert out.get(timeout=5) == 1


def test_two_rank_tie_replay_equals_reference_heap():
    _run(_tie_worker)


def test_three_rank_tie_replay_equals_reference_heap():
    _run(_tie_worker, world=3)


def test_two_rank_build_and_query_equal_single_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert out.get(timeout=5) == 1
