"""What one rank of an 8-GPU run does per batch, timed on ONE GPU: the bench index (10 M x 128,
PQ m=16 k=256) cut into `world` row shards, bounds of all shards computed once, then rank 0's
pipeline (scan_bounds -> [gathered bounds] -> scan_partial_bounded) timed against the plain
scan_partial, two batches in flight.   python tests/perf/bench_shared_bounds.py [world] [rows]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gulon_amd as g
from gulon_amd import native as N
from gulon_amd.recall import sample_rows
from gulon_amd.sharded import HipEngine, local_shard, shard_bounds

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
trace = len(sys.argv) > 3 and sys.argv[3] == "trace"      # one batch in flight, a few steps: for rocprofv3 timelines
d, m, k, B, K, nfl = 128, 16, 256, 1024, 10, (1 if trace else 2)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dm = g.DeviceMatrix.synthetic(n, d, 3, 1234, 1000)
pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, 10))
enc = pq.encode(dm)
Q = torch.from_numpy(dm.get_rows(sample_rows(n, B, 0))).to(dev)
abd = torch.empty((world * B, K + 1), dtype=torch.float32, device=dev)
lists = []
for r in range(world):
    lo, hi = shard_bounds(n, world, r)
    e = HipEngine(pq, local_shard(pq, enc, lo, hi), lo, dev)
    e.scan_bounds(Q, B, K, abd[r * B:(r + 1) * B])
    pv, pi = e.alloc((B, K + 1), "f32"), e.alloc((B, K + 1), "i32")
    e.scan_partial_bounded(Q, B, K, abd[r * B:(r + 1) * B], 1, pv, pi)     # own bound only: the reference lists
    lists.append((pv.clone(), pi.clone()))
    torch.cuda.synchronize()
    if r > 0:
        e.index.close()
    else:
        e0 = e
lo, hi = shard_bounds(n, world, 0)
engines = [e0] + [HipEngine(pq, local_shard(pq, enc, lo, hi), lo, dev) for _ in range(nfl - 1)]
streams = [torch.cuda.Stream() for _ in range(nfl)]
bufs = [(e.alloc((B, K + 1), "f32"), e.alloc((B, K + 1), "f32"), e.alloc((B, K + 1), "i32")) for e in engines]


def run(shared, steps=200):
    def step(i):
        e, (bd, pv, pi) = engines[i % nfl], bufs[i % nfl]
        with torch.cuda.stream(streams[i % nfl]):
            if shared:
                e.scan_bounds(Q, B, K, bd)
                e.scan_partial_bounded(Q, B, K, abd, world, pv, pi)
            else:
                e.scan_partial(Q, B, K, pv, pi)
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3


res = {"world": world, "rows_per_shard": hi - lo}
if trace:
    run(True, 6)
    sys.exit(0)
run(False); run(True)
res["own_bounds_ms"] = run(False)
res["shared_bounds_ms"] = run(True)
res["own_bounds_ms_again"] = run(False)
pv, pi = bufs[0][1].cpu().numpy(), bufs[0][2].cpu().numpy()
# with the global bound a shard's list may be shorter than with its own, but whatever can still reach the
# merged top-(K+1) is in it: merge all shards' lists both ways and compare
ref_v = np.stack([l[0].cpu().numpy() for l in lists]); ref_i = np.stack([l[1].cpu().numpy() for l in lists])
allv = np.concatenate(list(ref_v), axis=1); alli = np.concatenate(list(ref_i), axis=1)
order = np.lexsort((alli, allv), axis=1)[:, :K + 1]
top_v = np.take_along_axis(allv, order, 1); top_i = np.take_along_axis(alli, order, 1)
mine = (top_i >= lo) & (top_i < hi)
ok = all(set(top_i[q][mine[q]].tolist()) <= set(pi[q].tolist()) for q in range(B))
res["rank0_list_covers_global_topk"] = bool(ok)
for key in ("GULON_FILTER_SHARED_STAGE1",):
    for v in (0, 1):
        g.tune_live(**{key: v})
        res[f"shared_stage1_{v}_ms"] = run(True)
print(json.dumps(res))
