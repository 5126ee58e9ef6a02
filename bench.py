#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): queries/sec + recall@10 on a 10M x 128 PQ(m=16,k=256)
index, batch = 1024, K = 10, row-sharded over N GPUs (one process per GPU).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A step = one batch of 1024 queries through the hot path with everything resident in HBM:
distance-table build -> ADC scan of this rank's row shard + wavefront top-k -> (N > 1: RCCL
all-gather of the per-shard partial top-(K+1) lists) -> merge.  Index build (synthetic data,
PQ training, encoding) happens once, untimed, through the same library.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec


def kernel_source_sha16(kernel_name):
    """Identity of the kernel a committed counter pass describes: sha256 over the sources it is built from."""
    import hashlib
    files = {"filter_kernel": ["filter.hip", "conflict_order.hip", "scan.hpp", "common.hpp"],
             "scan_kernel": ["scan.hip", "scan.hpp", "common.hpp"]}[kernel_name]
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "gulon_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def run_steps(torch, shardeds, streams, Q, B, K, steps):
    """`steps` timed batches round-robin over the workspaces (after one untimed round); ms per step + last outputs."""
    nfl = len(shardeds)
    out = None
    for i in range(nfl):
        with torch.cuda.stream(streams[i]):
            shardeds[i].complete()
            out = shardeds[i].batch_query_dev(Q, B, K)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for s_ in range(steps):
        i = s_ % nfl
        with torch.cuda.stream(streams[i]):
            shardeds[i].complete()
            out = shardeds[i].batch_query_dev(Q, B, K)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / steps * 1e3
    return ms, tuple(x.cpu().numpy() for x in out)


def extras(args, g, N, L, torch, dev, HipEngine, ShardedIndex, local_shard, recall_at_k, sample_rows, dm, pq, shard,
           shardeds, streams, engines, Q, note):
    """Sub-records of the same JSON line (rank 0, one GPU): the cases that are NOT favourable to the pruning
    scan, and BASELINE config 3's k-means kernels, each over a few steps."""
    import ctypes as C
    n, d, m, k, B, K = args.rows, args.dim, args.quantizers, args.clusters, args.batch, args.knn
    steps = max(1, args.extra_steps)
    ex = {}

    def stats(engs, out):
        oi, od, oc, of = out
        tiles, redone = C.c_int32(0), C.c_int32(0)
        N.check(L.gulon_index_filter_stats(engs[-1 if len(engs) == 1 else (steps - 1) % len(engs)].index._h,
                                           C.byref(tiles), C.byref(redone)))
        return {"tie_flagged_queries": int(((of & 3) != 0).sum()), "replayed_queries": int(((of & 4) != 0).sum()),
                "query_tiles": tiles.value, "tiles_redone_by_exact_scan": redone.value}

    # 1. the exact scan (every look-up in fp32, no 8-bit filter): the path small ranges, wide indexes and the
    #    safety net take
    g.tune_live(GULON_SCAN_FILTER=0)     # every open handle and context (and the environment, for new ones)
    try:
        ms, _ = run_steps(torch, shardeds, streams, Q, B, K, min(steps, 3))
    finally:
        g.tune_live(GULON_SCAN_FILTER=1)
        os.environ.pop("GULON_SCAN_FILTER", None)
    ex["exact_scan"] = {"ms_per_step": ms, "queries_per_s": B / ms * 1e3,
                        "what": "GULON_SCAN_FILTER=0: same index and queries, every (query,row,quantizer) look-up in fp32"}
    note("extras: exact scan")
    # 2. held-out queries: rows n .. n+B of the same generator (same centres; none of them is in the index)
    big = g.DeviceMatrix.synthetic(n + B, d, args.data_kind, 1234, 1000)
    Qho_h = big.get_rows(np.arange(n, n + B, dtype=np.int32))
    del big
    Qho = torch.from_numpy(Qho_h).to(dev)
    ms, out = run_steps(torch, shardeds, streams, Qho, B, K, steps)
    rec = {"ms_per_step": ms, "queries_per_s": B / ms * 1e3,
           "what": f"queries = rows {n}..{n + B} of the same generator (not in the index)"}
    rec.update(stats(engines, out))
    if not args.no_recall:
        rec["recall_at_%d" % K], rec["recall_sd"] = recall_at_k(dm, Qho_h, K, out[0], out[2])
    ex["held_out_queries"] = rec
    note("extras: held-out queries")
    # 3. other data: kind 0 (iid N(0,1): PQ codes close to uniform, bounds separate least) and kind 1 (the
    #    reference-shaped generator: every row of a cluster shares its PQ code, every query ties)
    for kind in (0, 1):
        t0 = time.perf_counter()
        dmk = g.DeviceMatrix.synthetic(n, d, kind, 1234, 1000)
        pqk = g.ProductQuantizer.apply(dmk, g.ProductQuantizerConfig(k, m, args.train_iters))
        enck = pqk.encode(dmk)
        e0 = HipEngine(pqk, local_shard(pqk, enck, 0, n), 0, dev)
        engs = [e0] + [HipEngine(pqk, None, 0, dev, parent=e0) for _ in range(len(shardeds) - 1)]
        shs = [ShardedIndex(e, n, 0, 1, None, False) for e in engs]
        build_s = time.perf_counter() - t0
        qr = sample_rows(n, B, 0)
        Qk_h = dmk.get_rows(qr)
        Qk = torch.from_numpy(Qk_h).to(dev)
        ms, out = run_steps(torch, shs, streams, Qk, B, K, steps)
        rec = {"ms_per_step": ms, "queries_per_s": B / ms * 1e3, "build_seconds": build_s,
               "what": {0: "data kind 0: iid N(0,1) rows", 1: "data kind 1: 1000 tight clusters (the reference's generator shape): "
                        "rows of a cluster share their PQ code, every query is an exact tie"}[kind]}
        rec.update(stats(engs, out))
        if not args.no_recall:
            rec["recall_at_%d" % K], rec["recall_sd"] = recall_at_k(dmk, Qk_h, K, out[0], out[2])
        ex[f"data_kind_{kind}"] = rec
        for e in engs[1:] + engs[:1]:
            e.index.close()
        del shs, engs, e0, enck, pqk, dmk
        note(f"extras: data kind {kind}")
    # 4. BASELINE config 3: k-means codebook training, 10 M x 300, m = 32 (sub-dimensions 12 x 10 + 20 x 9), k = 256
    c3n, c3d, c3m, c3k, c3it = int(os.environ.get("GULON_BENCH_C3_ROWS", "10000000")), 300, 32, 256, 2
    dm3 = g.DeviceMatrix.synthetic(c3n, c3d, 2, 1234, 1)
    N.check(L.gulon_kmeans_trace(1))
    t0 = time.perf_counter()
    g.ProductQuantizer.apply(dm3, g.ProductQuantizerConfig(c3k, c3m, c3it))
    train_s = time.perf_counter() - t0
    tot = N.KMeansTraceTotals()
    N.check(L.gulon_kmeans_trace_read(C.byref(tot)))
    N.check(L.gulon_kmeans_trace(0))
    del dm3
    it = max(tot.iterations, 1)
    MFMA_F32_PEAK = 157.3     # MI355X_MICROARCH.md: fp32 matrix peak, TFLOP/s
    a_tf = tot.mfma_flops / (tot.assign_ms * 1e-3) / 1e12 if tot.assign_ms > 0 else 0.0
    u_gb = tot.update_bytes / (tot.update_ms * 1e-3) / 1e9 if tot.update_ms > 0 else 0.0
    ex["c3_kmeans"] = {
        "workload": f"{c3n}x{c3d} U[0,1), ProductQuantizer.apply m={c3m} k={c3k}, {it} iterations timed "
                    "(update -> parAssign -> convergence test), stages closed by device synchronisation",
        "train_seconds_total": train_s, "iterations": tot.iterations,
        "ms_per_iteration": {"update": tot.update_ms / it, "assign_mfma": tot.assign_ms / it,
                             "exact_recheck_and_tie_replay": tot.recheck_ms / it, "converged_test": tot.converge_ms / it},
        "assign_mfma": {"bound": "mfma", "achieved": a_tf, "peak": MFMA_F32_PEAK, "unit": "TFLOP/s",
                        "frac": a_tf / MFMA_F32_PEAK, "flops_per_iteration": tot.mfma_flops / it,
                        "rows_rechecked_exactly": tot.rows_rechecked / max(tot.rows_total, 1.0)},
        "update": {"bound": "hbm", "achieved": u_gb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": u_gb / HBM_PEAK_GBS,
                   "bytes_per_iteration": tot.update_bytes / it,
                   # what the streamed update moves: per sub-quantizer the pair-major slice once (8 bytes per
                   # row and dimension pair), the assignments (4 bytes per row), the chunk order written and read (2 + 2)
                   "physical": (lambda moved: {"bytes_moved_per_iteration": moved,
                                               "GB_per_s": moved / (tot.update_ms / it * 1e-3) / 1e9 if tot.update_ms > 0 else 0.0,
                                               "frac": moved / (tot.update_ms / it * 1e-3) / 1e9 / HBM_PEAK_GBS if tot.update_ms > 0 else 0.0})(
                       1.0 * c3n * sum(8 * ((s_ + 1) // 2) + 8 for s_ in
                                       [c3d // c3m + (1 if j < c3d % c3m else 0) for j in range(c3m)]))}}
    note("extras: C3 k-means")
    return ex


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=10_000_000)   # (not --n/--m/--d: torchrun would read them as its own)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--quantizers", type=int, default=16)
    ap.add_argument("--clusters", type=int, default=256)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--knn", type=int, default=10)
    ap.add_argument("--train-iters", type=int, default=10)
    ap.add_argument("--data-kind", type=int, default=3, help="0 iid, 1 clustered, 2 uniform, 3 overlapping clusters")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-recall", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the sub-records (exact scan, held-out queries, data kinds 0 and 1, C3 k-means)")
    ap.add_argument("--extra-steps", type=int, default=5, help="timed steps of every sub-record")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("GULON_BENCH_INFLIGHT", "0")),
                    help="query batches in flight (each on its own stream with its own scratch)")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON result: libraries that print to fd 1 on their own (RCCL's
    # version banner at communicator creation does) are sent to stderr instead
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import gulon_amd as g
    from gulon_amd import native as N
    from gulon_amd.recall import recall_at_k, sample_rows
    from gulon_amd.sharded import HipEngine, ShardedIndex, build_sharded, local_shard, shard_bounds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "GULON_BENCH_DEVICE" in os.environ:        # debugging aid: put every rank on one device
        local_rank = int(os.environ["GULON_BENCH_DEVICE"])
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    N.check(N.lib().gulon_set_device(local_rank))
    dist = None
    # GULON_BENCH_REHEARSE=1: one rank, the multi-rank pipeline with real RCCL calls over all --rows rows;
    # GULON_BENCH_REHEARSE=N > 1: this process is rank 0 of N -- 1/N of the rows, N lists per exchange, the other
    # ranks' slots of every gather prefilled once with what they send for the bench queries (ShardedIndex.prefill)
    rehearse = world == 1 and bool(os.environ.get("GULON_BENCH_REHEARSE"))
    emu = int(os.environ.get("GULON_BENCH_REHEARSE", "0") or 0) if rehearse else 0
    emu = emu if emu > 1 else 0
    if rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
    if world > 1 or rehearse:
        import torch.distributed as dist
        if os.environ.get("GULON_BENCH_BACKEND") == "gloo":   # rehearsal of the multi-rank path on one GPU
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            # the collectives' kernels queue for compute units like every other launch while a filter kernel
            # fills the chip: a high-priority stream gets them the units that come free first
            opts = None
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            except Exception:
                opts = None
            kw = {"pg_options": opts} if opts is not None else {}
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), **kw)
    dev = torch.device("cuda", local_rank)
    L = N.lib()
    n, d, m, k, B, K = args.rows, args.dim, args.quantizers, args.clusters, args.batch, args.knn
    t_origin = time.perf_counter()

    def note(what):     # progress on stderr (GULON_BENCH_VERBOSE=1): long builds are not mistaken for hangs
        if os.environ.get("GULON_BENCH_VERBOSE") and rank == 0:
            print(f"[bench +{time.perf_counter() - t_origin:7.2f}s] {what}", file=sys.stderr, flush=True)

    # ---- untimed build: synthetic clustered data -> PQ train -> encode (all on the GPU) -----
    t0 = time.perf_counter()
    dm = g.DeviceMatrix.synthetic(n, d, args.data_kind, 1234, 1000)
    t1 = time.perf_counter()
    note("synthetic data on the device")
    if world == 1:
        pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, args.train_iters))
        t2 = time.perf_counter()
        note("PQ trained")
        enc = pq.encode(dm)
        t3 = time.perf_counter()
        note("encoded")
        lo, hi = shard_bounds(n, emu if emu else world, rank)
        shard = local_shard(pq, enc, lo, hi)
    else:
        # quantizer-partitioned training + encoding, codebooks/codes all-gathered, rows re-sharded
        enc = None
        pq, shard, lo, hi = build_sharded(dm, k, m, args.train_iters, rank, world, dist,
                                          dev if dist.get_backend() == "nccl" else None)
        t2 = t3 = time.perf_counter()
    nloc = hi - lo
    coder = pq.coder_factory(nloc)
    # one engine (= device index + scratch) and one stream per batch in flight
    # batches in flight: 3 on one GPU (the others fill the gaps of one batch's short kernels: 2.477 / 2.443 / 2.432 ms
    # per step for 2 / 3 / 4 at 10 M rows -- the main-stage kernels run back to back from two on); 4 when
    # the shards exchange bounds, lists and replay candidates -- a batch waiting for its all-gathers (whose
    # kernels queue for compute units behind a running filter kernel like every other launch) must not leave
    # the GPU idle (1.25 M-row shard, one-rank RCCL rehearsal: 0.602 / 0.566 / 0.561 ms for 2 / 3 / 4)
    collective = world > 1 or rehearse
    nfl = args.inflight if args.inflight > 0 else (4 if collective else 3)
    # ONE copy of the shard's codes in HBM; every further batch in flight is a query context over it
    # (gulon_index_context_create: its own scratch, the same codes and codebooks)
    engines = [HipEngine(pq, shard, lo, dev)]
    engines += [HipEngine(pq, shard, lo, dev, parent=engines[0]) for _ in range(nfl - 1)]
    shardeds = [ShardedIndex(e, n, rank, world, dist, rehearse, emulate_world=emu) for e in engines]
    if collective:
        # side streams only: work on the legacy default stream serialises with the collectives' stream
        # (rehearsal, two batches in flight: 0.716 ms per step with the default stream among them, 0.615 without)
        streams = [torch.cuda.Stream() for _ in range(nfl)]
    else:
        streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nfl - 1)]
    engine, sharded, index = engines[0], shardeds[0], engines[0].index
    note("device index built")
    build_s = dict(synth=t1 - t0, train=t2 - t1, encode=t3 - t2)

    # ---- queries: B dataset rows drawn with java.util.Random(0) (Tests.scala:76-87) --------
    qrows = sample_rows(n, B, 0)
    Qh = dm.get_rows(qrows)
    Q = torch.from_numpy(Qh).to(dev)

    if emu:
        others = []
        for r in range(1, emu):
            rlo, rhi = shard_bounds(n, emu, r)
            others.append(HipEngine(pq, local_shard(pq, enc, rlo, rhi), rlo, dev))
        shardeds[0].prefill(Q, B, K, others)
        for s_ in shardeds[1:]:
            s_.copy_prefill(shardeds[0], B, K)
        for e in others:
            e.index.close()
        del others
        note(f"rank 0 of {emu}: the other ranks' slots prefilled")

    step_no = [0]

    def step():
        # table build -> local ADC scan + top-k -> (world > 1: all-gather partial lists) -> merge
        i = step_no[0] % nfl
        step_no[0] += 1
        with torch.cuda.stream(streams[i]):
            # the batch this slot ran `nfl` steps ago is consumed first (as a server hands results out): when it
            # had more tie-flagged queries than the unconditional replay round holds, its further rounds run here
            extra_rounds[0] += shardeds[i].complete()
            return shardeds[i].batch_query_dev(Q, B, K)

    extra_rounds = [0]

    def drain():
        for i in range(nfl):
            with torch.cuda.stream(streams[i]):
                extra_rounds[0] += shardeds[i].complete()

    for _ in range(nfl):
        out_idx, out_dist, out_cnt, out_flg = step()
    drain()
    torch.cuda.synchronize()
    step_no[0] = 0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    note("first batches done")
    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    extra_rounds[0] = 0
    for e in engines:
        N.check(L.gulon_index_profile(e.index._h, 1))
    barrier()
    if os.environ.get("GULON_HOST_PROFILE"):
        ShardedIndex.host_profile = {"_last": time.perf_counter()}
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    enqueue_s = time.perf_counter() - t_start          # host time to enqueue K steps (diagnostic)
    if ShardedIndex.host_profile is not None and rank == 0:
        hp, ShardedIndex.host_profile = ShardedIndex.host_profile, None
        for name, sec in sorted(hp.items(), key=lambda kv: -kv[1] if kv[0] != "_last" else 0):
            if name != "_last":
                print(f"[host] {sec / args.steps * 1e6:8.1f} us/step  {name}", file=sys.stderr)
    barrier()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    ms_total, launches, rows_cov = C.c_double(0), C.c_int32(0), C.c_int64(0)
    for e in engines:
        a, b_, c_ = C.c_double(0), C.c_int32(0), C.c_int64(0)
        N.check(L.gulon_index_profile_read_ex(e.index._h, C.byref(a), C.byref(b_), C.byref(c_)))
        N.check(L.gulon_index_profile(e.index._h, 0))
        ms_total.value += a.value; launches.value += b_.value; rows_cov.value += c_.value
    # dominant kernel: the main-stage launch of the quantized filter when it is active (one launch per
    # batch over ~95 % of the rows), else the exact scan; averages are per launch, like rocprofv3 --stats
    scan_ms = ms_total.value / max(launches.value, 1)
    rows_per_launch = rows_cov.value / max(launches.value, 1)
    filtered = rows_per_launch < 0.999 * nloc
    kernel_name = "filter_kernel" if filtered else "scan_kernel"

    note("timed region done")
    ms_per_step = elapsed / args.steps * 1e3
    qps = B * args.steps / elapsed
    alg_bytes = float(B) * rows_per_launch * m            # m code bytes per (query, row) pair
    achieved = alg_bytes / (scan_ms * 1e-3) / 1e9 if scan_ms > 0 else 0.0
    # physical side of the same kernel: HBM bytes and LDS / VALU counters per launch come from the committed
    # rocprofv3 --pmc passes of this very command (profiles/scan_traffic.json names the csv they were read from);
    # the duration they are divided by is the one measured live above
    traffic, physical, physical_note = None, None, None
    tf = os.path.join(ROOT, "profiles", "scan_traffic.json")
    if os.path.exists(tf):
        try:
            rec = json.load(open(tf)).get(f"{kernel_name}_n{nloc}_m{m}_B{B}")
            if rec is None:
                physical_note = f"no counter pass committed for {kernel_name} at n={nloc}, m={m}, B={B}"
            elif rec.get("source_sha16") != kernel_source_sha16(kernel_name):
                # counters of another version of the kernel say nothing about this one
                physical_note = (f"stale: the committed counters ({rec.get('source')}) were taken from kernel sources "
                                 f"{rec.get('source_sha16')}, this build is {kernel_source_sha16(kernel_name)}")
                rec = None
            traffic = rec["hbm_bytes_per_launch"] if rec else None
            if rec and scan_ms > 0:
                hbm_gbs = traffic / (scan_ms * 1e-3) / 1e9
                physical = {"hbm": {"bytes_per_launch": traffic, "GB_per_s": hbm_gbs, "peak_GB_per_s": HBM_PEAK_GBS,
                                    "frac": hbm_gbs / HBM_PEAK_GBS,
                                    "traffic_over_algorithmic": traffic / alg_bytes if alg_bytes else None},
                            "source": rec.get("source")}
                if "lds_idx_active" in rec:
                    cyc = rec["grbm_gui_active"] / 8.0                       # kernel cycles (sum over 8 XCDs / 8)
                    free = (rec["lds_idx_active"] - rec["lds_bank_conflict"]) / rec.get("cus", 256)
                    physical["lds"] = {
                        "bound": "LDS gather (ds_read_b128 of 16 one-byte bounds per look-up)",
                        "conflict_free_cycles_per_cu": free, "kernel_cycles": cyc, "frac": free / cyc,
                        "pipe_busy": rec["lds_idx_active"] / rec.get("cus", 256) / cyc,
                        "bank_conflict_share": rec["lds_bank_conflict"] / rec["lds_idx_active"]}
                if "valu_busy" in rec:
                    physical["valu_busy"] = rec["valu_busy"]
        except Exception:
            traffic, physical = None, None

    res_idx = out_idx.cpu().numpy()
    res_cnt = out_cnt.cpu().numpy()
    res_dist = out_dist.cpu().numpy()
    res_flg = out_flg.cpu().numpy()

    result = {
        "metric": "queries_per_sec", "value": qps, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{n}x{d} synthetic (kind {args.data_kind}, 1000 centres, seed 1234), PQ(m={m},k={k}) flat ADC scan, batch={B}, K={K}, "
                               f"rows sharded over {world} GPU(s)", "n": n, "d": d, "m": m, "k": k, "batch": B,
                   "knn": K, "train_max_iterations": args.train_iters, "rows_per_gpu": nloc,
                   "batches_in_flight": nfl,
                   "dist_backend": dist.get_backend() if dist is not None else None,
                   "dist_world_size": dist.get_world_size() if dist is not None else 1,
                   "tie_replay_extra_rounds": extra_rounds[0]},
        # contract fields: ALGORITHMIC bytes (m code bytes per (query, row) pair) over the kernel's
        # duration against the HBM peak.  The kernel reads every code word once per 16-query tile out of
        # L2 / Infinity Cache and decides > 99.9 % of the pairs on 8-bit bounds, so this "fraction" exceeds 1 and
        # bounds nothing; `physical` holds the bounds that bind (HBM actually moved; the LDS gather pipe)
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kernel_name,
                     "kernel_ms": scan_ms, "launches_per_step": launches.value / max(args.steps, 1),
                     "algorithmic_bytes_per_launch": alg_bytes, "basis": "algorithmic bytes, not DRAM traffic",
                     "physical": physical, "physical_note": physical_note},
        "build_seconds": build_s,
        "host_enqueue_ms_per_step": enqueue_s / args.steps * 1e3,
    }
    if rehearse:
        result["config"]["rehearsal"] = (
            f"rank 0 of {emu}: {nloc} of the {n} rows, {emu} lists per exchange (RCCL all-gathers over a one-rank group "
            f"move this rank's slot; the other ranks' bounds, lists and tie candidates for these queries were computed "
            f"once from their shards)" if emu else
            "one rank through the multi-rank pipeline (RCCL all-gathers of one list)")

    if rank == 0:
        if not args.no_recall:
            t = time.perf_counter()
            mean, sd = recall_at_k(dm, Qh, K, res_idx, res_cnt)
            result["recall_at_10" if K == 10 else f"recall_at_{K}"] = mean
            result["recall_sd"] = sd
            result["recall_seconds"] = time.perf_counter() - t
            result["tie_flagged_queries"] = int((res_flg != 0).sum())
            note("recall done")
        if world == 1 and enc is not None and not args.no_extras and not rehearse:
            result["extras"] = extras(args, g, N, L, torch, dev, HipEngine, ShardedIndex, local_shard, recall_at_k,
                                      sample_rows, dm, pq, shard, shardeds, streams, engines, Q, note)
        if world == 1 and enc is not None and not args.no_cpu_baseline:
            from oracle import oracle                      # CPU baseline leg only (the checker, timed)
            codes_h = np.stack(enc.encodings) if coder.width == 8 else None
            cents = pq.flat_centroids()
            if codes_h is not None:
                cq, spent, t_cpu, n_q = 1, 0.0, 0.0, 0
                oi = od = None
                ois, ods = [], []
                while spent < args.cpu_seconds and n_q < B:
                    cq = min(cq, B - n_q)
                    t = time.perf_counter()
                    a, b_, _ = oracle.pq_batch_query(codes_h, d, k, cents, Qh[n_q:n_q + cq], K)
                    dt = time.perf_counter() - t
                    ois.append(a); ods.append(b_)
                    spent += dt; t_cpu += dt; n_q += cq
                    per_q = t_cpu / n_q
                    cq = max(1, min(int((args.cpu_seconds - spent) / per_q), 64))
                    if args.cpu_seconds - spent < per_q:
                        break
                oi, od = np.concatenate(ois), np.concatenate(ods)
                same_d = bool(np.array_equal(od.view(np.uint32), res_dist[:n_q].view(np.uint32)))
                # ids are claimed exact for unflagged queries AND for replayed ones (flag 4); a flagged query
                # that was not replayed may differ only inside the tie group of its last distance
                def ids_ok(q):
                    if res_flg[q] == 0 or (res_flg[q] & 4):
                        return np.array_equal(oi[q], res_idx[q])
                    inner = od[q] < od[q][-1]
                    return set(oi[q][inner].tolist()) == set(res_idx[q][inner].tolist())
                same_i = bool(all(ids_ok(q) for q in range(n_q)))
                result["cpu_baseline"] = {
                    "value": n_q / t_cpu, "unit": "queries/s", "cores": 1, "kind": "port",
                    "sample": f"first {n_q} of the {B} queries over all {n} rows, single thread, "
                              f"4096-row blocks (Index.scala:424); C restatement of Gulon's JVM algorithm"}
                result["parity_vs_oracle"] = {
                    "queries": n_q, "distances_bit_exact": same_d, "ids_equal": same_i,
                    "tie_flagged": int((res_flg[:n_q] != 0).sum()), "replayed": int(((res_flg[:n_q] & 4) != 0).sum()),
                    "flagged_not_replayed": int(((res_flg[:n_q] != 0) & ((res_flg[:n_q] & 4) == 0)).sum())}
                # the same restatement with the queries striped over the host cores -- one index.query per
                # task, as Tests.recallOf runs them (Tests.scala:22-29, parTraverse); a second, bounded sample
                cores = min(len(os.sched_getaffinity(0)), 16)
                if cores > 1 and args.cpu_seconds > 0:
                    from concurrent.futures import ThreadPoolExecutor
                    per_thread = max(1, min(int(0.6 * args.cpu_seconds * (n_q / t_cpu)), B // cores))
                    sl = [Qh[i * per_thread:(i + 1) * per_thread] for i in range(cores)]
                    t = time.perf_counter()
                    with ThreadPoolExecutor(cores) as ex:       # ctypes releases the GIL inside the C scan
                        outs = list(ex.map(lambda q: oracle.pq_batch_query(codes_h, d, k, cents, q, K), sl))
                    dt = time.perf_counter() - t
                    par_d = np.concatenate([o[1] for o in outs])
                    result["cpu_baseline_all_cores"] = {
                        "value": cores * per_thread / dt, "unit": "queries/s", "cores": cores, "kind": "port",
                        "sample": f"first {cores * per_thread} queries, {per_thread} per thread, all {n} rows",
                        "distances_bit_exact": bool(np.array_equal(par_d.view(np.uint32),
                                                                   res_dist[:cores * per_thread].view(np.uint32)))}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
