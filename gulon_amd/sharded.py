"""Row-sharded PQIndex over the GPUs of one node: one process per GPU.

The code matrix is split into `world` contiguous row ranges -- exactly the
from/until contract of PQIndex.batchQuery (Index.scala:417-419).  Every rank
builds the same distance tables, scans its own rows, and the per-shard partial
top-(K+1) lists are exchanged with ONE all-gather (RCCL over xGMI when
the process group is "nccl"); each rank then merges the `world` lists with
TopKHeap.merge semantics (TopKHeap.scala:44-53, used the same way by
Index.scala:279) under the deterministic (distance, row id) order, so the
result does not depend on the number of shards.

Queries flagged with an exact distance tie are then replayed with the reference's heap
semantics from the candidate rows of all shards: a fixed-size all-gather for the first 16 flagged
queries, enqueued unconditionally, and -- only for batches with more of them (`complete()`) -- further
rounds of 128 until every flagged query has been replayed, so that ids and order equal the unsharded
index's whatever the number of shards and of ties.

The compute engine is pluggable only so that the orchestration (bounds, row
bases, gather layout) can be exercised on CPU with the gloo backend in tests;
the shipped engine is HipEngine (libgulon_hip.so) and nothing else.
"""
import ctypes as C

import numpy as np

from . import native as N
from .product_quantizer import EncodedMatrix, ProductQuantizer


def shard_bounds(n, world, rank):
    """Rows [lo, hi) owned by `rank`: balanced contiguous ranges."""
    return n * rank // world, n * (rank + 1) // world


def local_shard(pq: ProductQuantizer, encoded: EncodedMatrix, lo, hi) -> EncodedMatrix:
    """Slice a full EncodedMatrix (every encodings(j) covers all n rows) to rows [lo, hi)."""
    coder = pq.coder_factory(hi - lo)
    if coder.width == 8:
        return EncodedMatrix(coder, [e[lo:hi] for e in encoded.encodings])
    idx = encoded.indices()
    return EncodedMatrix(coder, [coder.build_code(idx[j, lo:hi]) for j in range(idx.shape[0])])


class HipEngine:
    """Local scan + merge on this rank's GPU through the C ABI (device-resident)."""

    def __init__(self, pq, shard: EncodedMatrix, row_base, device, parent=None):
        import torch
        from .index import PQIndex
        # torch's HIP runtime has to come up BEFORE libgulon_hip.so touches the device in this process
        # (the other order leaves torch with "No HIP GPUs are available"): bench.py calls
        # torch.cuda.set_device first; do the same in any process that mixes the two.
        if not torch.cuda.is_initialized():
            torch.cuda.init()
        self.torch = torch
        self.device = device
        # parent: another HipEngine over the same shard -- this engine is then only a WORKSPACE (a query context,
        # gulon_index_context_create) over that engine's codes: one per batch in flight, no second copy in HBM
        self.index = PQIndex(pq, shard, row_base=row_base) if parent is None else parent.index.context()
        self.nloc = shard.length if shard is not None else parent.nloc
        # flagged queries per round of the exact tie replay: the first, unconditional round is sized for the
        # common case (a few ties per batch), later rounds (only when a batch needs them) take more at once
        self.replay_first, self.replay_more, self.replay_pool = N.REPLAY_MAX_FLAGGED, 128, N.REPLAY_POOL

    def replay_words(self, flagged):
        """int32 words of one shard's candidate buffer for `flagged` tie-flagged queries."""
        return int(N.lib().gulon_replay_pack_words(flagged, self.replay_pool))

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def alloc(self, shape, dtype):
        t = self.torch
        return t.empty(shape, dtype={"f32": t.float32, "i32": t.int32}[dtype], device=self.device)

    def to_device(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def query_final(self, q, b, k, oi, od, oc, of):
        N.check(N.lib().gulon_index_batch_query_dev(self.index._h, q.data_ptr(), b, k, 0, self.nloc, oi.data_ptr(),
                                                    od.data_ptr(), oc.data_ptr(), of.data_ptr(), self._stream()))

    def scan_partial(self, q, b, k, pv, pi):
        N.check(N.lib().gulon_index_scan_partial_dev(self.index._h, q.data_ptr(), b, k, 0, self.nloc, pv.data_ptr(),
                                                     pi.data_ptr(), self._stream()))

    def scan_bounds(self, q, b, k, bd):
        """First half of scan_partial: this shard's K+1 smallest sample distances per query -> bd [B][K+1]."""
        N.check(N.lib().gulon_index_scan_bounds_dev(self.index._h, q.data_ptr(), b, k, 0, self.nloc, bd.data_ptr(),
                                                    self._stream()))

    def scan_partial_bounded(self, q, b, k, abd, lists, pv, pi):
        """Second half: the scan against the bound of the union of all shards' samples (abd [lists][B][K+1])."""
        N.check(N.lib().gulon_index_scan_partial_bounded_dev(self.index._h, q.data_ptr(), b, k, 0, self.nloc,
                                                             abd.data_ptr(), lists, pv.data_ptr(), pi.data_ptr(),
                                                             self._stream()))

    def merge(self, packed, lists, b, k, oi, od, oc, of):
        """packed: [lists][2][B][K+1] int32 words (distance bits, then row ids) as gathered."""
        base = packed.data_ptr()
        N.check(N.lib().gulon_topk_merge_dev(base, base + 4 * b * (k + 1), lists, 2 * b * (k + 1), b, k,
                                             oi.data_ptr(), od.data_ptr(), oc.data_ptr(), of.data_ptr(),
                                             self._stream()))

    def views(self, pk, b):
        """(float32 view of the distance half, int32 row-id half) of a [2*B][K+1] int32 buffer."""
        return pk[:b].view(self.torch.float32), pk[b:]

    def replay_collect(self, q, b, k, of, pack, skip, flagged):
        """This shard's candidate rows for the TopKHeap replay of flagged queries [skip, skip + flagged)."""
        N.check(N.lib().gulon_index_replay_collect_dev(self.index._h, q.data_ptr(), b, k, 0, self.nloc, of.data_ptr(),
                                                       skip, flagged, self.replay_pool, pack.data_ptr(), self._stream()))

    def replay_apply(self, packs, lists, b, k, oi, od, oc, of, flagged):
        N.check(N.lib().gulon_replay_apply_dev(packs.data_ptr(), lists, flagged, self.replay_pool, b, k, oi.data_ptr(),
                                               od.data_ptr(), oc.data_ptr(), of.data_ptr(), self._stream()))

    def replay_total(self, pack):
        """Flagged queries of the whole batch (word 3 of a collected pack); waits for the pack."""
        return int(pack[3].item())

    def nan_fix(self, q, b, k, n_total, oi, od, oc, of):
        """All-NaN queries: the reference's first-K-rows answer (gulon_nan_queries_fix_dev)."""
        N.check(N.lib().gulon_nan_queries_fix_dev(q.data_ptr(), b, q.shape[1], k, n_total, oi.data_ptr(), od.data_ptr(),
                                                  oc.data_ptr(), of.data_ptr(), self._stream()))


class ShardedIndex:
    """One rank's view of the row-sharded flat index."""

    def __init__(self, engine, n_total, rank=0, world=1, dist=None, rehearse=False, emulate_world=0):
        self.engine = engine
        self.n_total, self.rank, self.world, self.dist = n_total, rank, world, dist
        # emulate_world = N > 1 (bench.py, GULON_BENCH_REHEARSE=N on a one-GPU box): this process does exactly what
        # rank 0 of N does per batch -- its 1/N of the rows, N lists per exchange -- over a ONE-rank process group:
        # every all-gather moves this rank's slot, the other N - 1 slots hold what the other ranks would have sent
        # for the same queries (`prefill`, computed once from their shards)
        self.emulate = emulate_world > 1
        if self.emulate:
            self.world = emulate_world
        # the multi-rank pipeline (partial lists -> all-gather -> merge -> replay exchange); `rehearse`
        # runs it with a one-rank process group as well, so that the RCCL path can be exercised and
        # timed on a one-GPU box (bench.py: GULON_BENCH_REHEARSE=1)
        self.collective = world > 1 or (rehearse and dist is not None)
        # shards share their pruning bounds (one more, tiny all-gather in front of the scan) when the
        # engine can split its scan; GULON_SHARED_BOUNDS=0 keeps every shard on its own bound
        import os
        self.share_bounds = (self.collective and hasattr(engine, "scan_bounds")
                             and os.environ.get("GULON_SHARED_BOUNDS", "1") != "0")
        self.lo, self.hi = shard_bounds(n_total, self.world, rank)
        self._bufs = {}
        # RCCL gathers device tensors directly; a gloo group (CPU rehearsal of the multi-rank path
        # with the real HIP engine) needs the lists staged through the host
        self.host_staged = bool(dist is not None and self.collective and dist.get_backend() == "gloo"
                                and getattr(engine, "device", None) is not None)

    def _buffers(self, b, k):
        key = (b, k)
        if key not in self._bufs:
            e = self.engine
            self._bufs[key] = dict(
                oi=e.alloc((b, k), "i32"), od=e.alloc((b, k), "f32"), oc=e.alloc((b,), "i32"),
                of=e.alloc((b,), "i32"),
                # this rank's partial list, one buffer so that ONE all-gather moves it:
                # rows [0, B) = distance bits, rows [B, 2B) = row ids
                pk=e.alloc((2 * b, k + 1), "i32"),
                # gathered, rank-major: [world][2][B][K+1]
                apk=e.alloc((self.world * 2 * b, k + 1), "i32"))
            if self.share_bounds:
                self._bufs[key]["bd"] = e.alloc((b, k + 1), "f32")
                self._bufs[key]["abd"] = e.alloc((self.world * b, k + 1), "f32")
            if hasattr(e, "replay_words") and self.collective:
                words = e.replay_words(e.replay_first)
                self._bufs[key]["rp"] = e.alloc((words,), "i32")
                self._bufs[key]["arp"] = e.alloc((self.world * words,), "i32")
        return self._bufs[key]

    def _all_gather(self, out, inp):
        if self.emulate:
            # one-rank group: the collective writes slot 0; slots 1 .. N-1 were filled by prefill()
            self.dist.all_gather_into_tensor(out.view(-1)[:inp.numel()].view(inp.shape), inp)
            return
        if self.host_staged:
            # rehearsal path (gloo has no device collectives): same layout, staged through the host
            hi, ho = inp.cpu(), out.cpu()
            self.dist.all_gather_into_tensor(ho, hi)
            out.copy_(ho)
        else:
            self.dist.all_gather_into_tensor(out, inp)

    def batch_query_dev(self, q, b, k):
        """Enqueue one batch; returns the (device) output tensors idx, dist, count, flags."""
        u = self._buffers(b, k)
        tick = self._tick
        if not self.collective:
            self.engine.query_final(q, b, k, u["oi"], u["od"], u["oc"], u["of"])
        else:
            pv, pi = self.engine.views(u["pk"], b)
            large_k = k > N.MAX_K      # beyond a wavefront list: peeled partial lists, long merge, (distance, row) ties
            if self.share_bounds and not large_k:
                self.engine.scan_bounds(q, b, k, u["bd"]); tick("scan_bounds")
                self._all_gather(u["abd"], u["bd"]); tick("all_gather bounds")
                self.engine.scan_partial_bounded(q, b, k, u["abd"], self.world, pv, pi); tick("scan_partial_bounded")
            else:
                self.engine.scan_partial(q, b, k, pv, pi); tick("scan_partial")
            self._all_gather(u["apk"], u["pk"]); tick("all_gather lists")
            self.engine.merge(u["apk"], self.world, b, k, u["oi"], u["od"], u["oc"], u["of"]); tick("merge")
            if "rp" in u and not large_k:
                # queries with exact distance ties: every shard contributes the rows that may insert into
                # the reference's heap, the union is replayed identically on every rank (TopKHeap.scala:57-79).
                # This first round (replay_first queries) is unconditional -- no host synchronisation;
                # complete() looks at the batch's flag count afterwards and runs further rounds if needed.
                e = self.engine
                e.replay_collect(q, b, k, u["of"], u["rp"], 0, e.replay_first); tick("replay_collect")
                self._all_gather(u["arp"], u["rp"]); tick("all_gather candidates")
                e.replay_apply(u["arp"], self.world, b, k, u["oi"], u["od"], u["oc"], u["of"], e.replay_first); tick("replay_apply")
                self._pending = (q, b, k)
            if hasattr(self.engine, "nan_fix"):
                self.engine.nan_fix(q, b, k, self.n_total, u["oi"], u["od"], u["oc"], u["of"]); tick("nan_fix")
        return u["oi"], u["od"], u["oc"], u["of"]

    # host-side cost of a step, call by call (GULON_HOST_PROFILE=1; bench.py prints the table to stderr)
    host_profile = None

    def _tick(self, name):
        hp = ShardedIndex.host_profile
        if hp is not None:
            import time
            now = time.perf_counter()
            hp[name] = hp.get(name, 0.0) + now - hp["_last"]
            hp["_last"] = now

    def prefill(self, q, b, k, other_engines):
        """Emulation of rank 0 of N: fill slots 1 .. N-1 of every gather buffer with what ranks 1 .. N-1 send for the
        batch (q, b, k) -- their sample bounds, their partial lists against the shared bound, their tie-replay
        candidates.  other_engines[r - 1] = an engine over rank r's rows (row_base = its first row)."""
        assert self.emulate and self.share_bounds and len(other_engines) == self.world - 1
        u, W = self._buffers(b, k), self.world
        engs = [self.engine] + list(other_engines)
        abd = u["abd"].view(W, b, k + 1)
        for r, e in enumerate(engs):
            e.scan_bounds(q, b, k, abd[r])       # (left pending; the next call on the handle drops it)
        apk = u["apk"].view(W, 2 * b, k + 1)
        for r, e in enumerate(engs):
            e.scan_bounds(q, b, k, u["bd"])
            pv, pi = e.views(apk[r], b)
            e.scan_partial_bounded(q, b, k, u["abd"], W, pv, pi)
        self.engine.merge(u["apk"], W, b, k, u["oi"], u["od"], u["oc"], u["of"])
        if "rp" in u:
            arp = u["arp"].view(W, -1)
            for r, e in enumerate(engs):
                e.replay_collect(q, b, k, u["of"], arp[r], 0, self.engine.replay_first)
        self.engine.torch.cuda.synchronize()

    def copy_prefill(self, other, b, k):
        """The same slots 1 .. N-1 for another workspace of the same shard (one per batch in flight)."""
        u, v = self._buffers(b, k), other._buffers(b, k)
        for key in ("abd", "apk", "arp"):
            if key in u:
                u[key].copy_(v[key])

    def complete(self):
        """Finish the batch enqueued last: when more queries were tie-flagged than the first replay round
        holds, run further rounds (collect -> all-gather -> literal TopKHeap) until every flagged query has
        been replayed.  Waits for the batch (reads its flag count); every rank sees the same count, so all
        ranks run the same number of rounds.  Returns the number of extra rounds."""
        pend, self._pending = getattr(self, "_pending", None), None
        if ShardedIndex.host_profile is not None:
            import time
            ShardedIndex.host_profile["_last"] = time.perf_counter()
        if pend is None:
            return 0
        q, b, k = pend
        u, e = self._buffers(b, k), self.engine
        total, skip, rounds = e.replay_total(u["rp"]), e.replay_first, 0
        self._tick("complete: wait for the flag count of this slot's previous batch")
        self.last_flagged = total
        if self.emulate and total > skip:
            raise RuntimeError(f"rank emulation holds the other ranks' candidates for {skip} tie-flagged queries only; "
                               f"this batch has {total}")
        if total > skip:
            words = e.replay_words(e.replay_more)
            if "rp2" not in u:
                u["rp2"] = e.alloc((words,), "i32")
                u["arp2"] = e.alloc((self.world * words,), "i32")
            while skip < total:
                e.replay_collect(q, b, k, u["of"], u["rp2"], skip, e.replay_more)
                self._all_gather(u["arp2"], u["rp2"])
                e.replay_apply(u["arp2"], self.world, b, k, u["oi"], u["od"], u["oc"], u["of"], e.replay_more)
                skip += e.replay_more
                rounds += 1
        return rounds

    def batch_query(self, k, queries):
        """Host convenience: numpy in, numpy out (idx [B][K], dist, count, flags)."""
        q = N.f32(queries)
        b = q.shape[0]
        oi, od, oc, of = self.batch_query_dev(self.engine.to_device(q), b, k)
        self.complete()
        return tuple(t.cpu().numpy() for t in (oi, od, oc, of))


# ---------------------------------------------------------------------------------------
# Build side: the m sub-quantizers are independent k-means problems
# (ProductQuantizer.scala:130-145), so they are partitioned over the ranks -- no collective
# inside the training loop, which keeps the order-dependent fp32 running means bit-exact.
# One all-gather of the codebooks at the end, one all-gather of the per-quantizer code
# arrays, then every rank keeps its row range.
# ---------------------------------------------------------------------------------------
def _all_gather_np(arr, dist, device):
    """All-gather equally-shaped numpy arrays -> [world, ...] numpy (device tensors under RCCL)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(arr))
    use_dev = device is not None and dist.get_backend() == "nccl"
    if use_dev:
        t = t.to(device)
    out = torch.empty((dist.get_world_size(),) + tuple(t.shape), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out.view(-1, *t.shape[1:]) if t.dim() > 1 else out.view(-1), t)
    return out.cpu().numpy()


def build_sharded(dm, num_clusters, num_quantizers, max_iterations, rank=0, world=1, dist=None, device=None):
    """ProductQuantizer.apply + encode with the quantizers partitioned over `world` ranks.

    Returns (pq, shard, lo, hi): the full ProductQuantizer (identical on every rank and to the
    single-process result) and the EncodedMatrix of this rank's row range [lo, hi)."""
    import ctypes as C
    from .vectors import subvector_bounds
    n, d, m, k = dm.rows, dm.cols, num_quantizers, num_clusters
    fr, un = subvector_bounds(d, m)
    jlo, jhi = shard_bounds(m, world, rank)
    cents = np.zeros(k * d, np.float32)
    N.check(N.lib().gulon_pq_train_range(dm._h, m, k, max_iterations, jlo, jhi, cents, None, 0, None))
    if world > 1:
        allc = _all_gather_np(cents, dist, device)                      # [world][k*d]
        for r in range(world):
            a, b = shard_bounds(m, world, r)
            if b > a:
                cents[k * fr[a]: k * un[b - 1]] = allc[r, k * fr[a]: k * un[b - 1]]
    pq = ProductQuantizer.from_flat(k, d, m, cents)

    coder = pq.coder_factory(n)
    bpc = coder.bytes_per_code
    mmax = max(shard_bounds(m, world, r)[1] - shard_bounds(m, world, r)[0] for r in range(world))
    local = np.zeros((mmax, max(bpc, 1)), np.uint8)
    if jhi > jlo and bpc > 0:
        buf = np.zeros((jhi - jlo) * bpc, np.uint8)
        N.check(N.lib().gulon_pq_encode_range(dm._h, m, k, cents, jlo, jhi, buf))
        local[: jhi - jlo, :bpc] = buf.reshape(jhi - jlo, bpc)
    if world > 1:
        allcodes = _all_gather_np(local, dist, device)                  # [world][mmax][bpc]
        encs = []
        for r in range(world):
            a, b = shard_bounds(m, world, r)
            encs.extend(allcodes[r, j, :bpc] for j in range(b - a))
    else:
        encs = [local[j, :bpc] for j in range(m)]
    full = EncodedMatrix(coder, encs)
    lo, hi = shard_bounds(n, world, rank)
    return pq, local_shard(pq, full, lo, hi), lo, hi


# ---------------------------------------------------------------------------------------
# One process, several GPUs: the C ABI's own sharded index (what a JVM binds).
# ---------------------------------------------------------------------------------------
class NodeShardedIndex:
    """PQIndex row-sharded over the GPUs of this node inside ONE process (gulon_sharded_index_*,
    sharded.hip): shard s holds rows [n*s/S, n*(s+1)/S) on devices[s]; bounds, partial lists and
    tie-replay candidates travel between the devices by RCCL all-gathers issued from this process.
    Results equal PQIndex.batch_query bit for bit (Index.scala:417-440, TopKHeap.scala:44-79)."""

    def __init__(self, product_quantizer: ProductQuantizer, data: EncodedMatrix, devices):
        self.product_quantizer, self.data = product_quantizer, data
        dev = N.i32(list(devices))
        h = C.c_void_p()
        packed = data.packed()
        N.check(N.lib().gulon_sharded_index_create(packed if packed.size else np.zeros(1, np.uint8), data.length,
                                                   product_quantizer.dimension, len(product_quantizer.quantizers),
                                                   product_quantizer.num_clusters, product_quantizer.flat_centroids(),
                                                   dev, len(dev), C.byref(h)))
        self._h = h

    def info(self):
        v = [C.c_int32(0) for _ in range(5)]
        N.check(N.lib().gulon_sharded_index_info(self._h, *[C.byref(x) for x in v]))
        return dict(shards=v[0].value, devices=v[1].value, rccl_version=v[2].value, last_replay_rounds=v[3].value,
                    last_flagged_queries=v[4].value)

    def batch_query_raw(self, k, vectors):
        q = N.f32(vectors).reshape(-1, self.product_quantizer.dimension)
        b = q.shape[0]
        oi = np.zeros((b, max(k, 1)), np.int32)
        od = np.zeros((b, max(k, 1)), np.float32)
        oc = np.zeros(max(b, 1), np.int32)
        of = np.zeros(max(b, 1), np.int32)
        N.check(N.lib().gulon_sharded_index_batch_query(self._h, q.reshape(-1) if b else np.zeros(1, np.float32), b, k,
                                                        oi.reshape(-1), od.reshape(-1), oc, of))
        return oi[:, :k], od[:, :k], oc[:b], of[:b]

    def batch_query(self, k, vectors):
        from .index import Result
        oi, od, oc, of = self.batch_query_raw(k, vectors)
        return [Result(oi[i, :oc[i]].copy(), od[i, :oc[i]].copy(), int(of[i])) for i in range(len(oc))]

    def close(self):
        if self._h is not None and self._h.value:
            N.lib().gulon_sharded_index_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
