"""GroupedIndex benchmark: coarse KMeans -> grouping -> residual PQ -> batched queries
with the reference CLI's defaults (partitions = n / 1000, LimitGroups(max(5 % of partitions, 5)),
BuildIndex.scala:104-106).   python tests/perf/bench_grouped.py [rows] [dim] [partitions] [limit]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gulon_amd as g
from gulon_amd import native as N
from gulon_amd.recall import recall_at_k, sample_rows

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
parts = int(sys.argv[3]) if len(sys.argv) > 3 else n // 1000
limit = int(sys.argv[4]) if len(sys.argv) > 4 else max(int(parts * 0.05), 5)
m, k, B, K, iters = 16, 256, 1024, 10, 10
L = N.lib()
t0 = time.perf_counter()
dm = g.DeviceMatrix.synthetic(n, d, 3, 1234, 1000)
t1 = time.perf_counter()
coarse = g.KMeans.compute_clusters(g.Vectors(dm), g.KMeansConfig(parts, iters))
t2 = time.perf_counter()
gv = g.group(dm, coarse)
t3 = time.perf_counter()
pq = g.ProductQuantizer.apply(gv.residuals, g.ProductQuantizerConfig(k, m, iters))
t4 = time.perf_counter()
index = g.Index.grouped(gv, pq, g.LimitGroups(limit))
t5 = time.perf_counter()
print(f"[grouped] n={n} d={d} partitions={parts} (non-empty {len(gv.centroids)}) limit={limit}: synth {t1-t0:.2f}s "
      f"coarse k-means {t2-t1:.2f}s group+residuals {t3-t2:.2f}s residual PQ {t4-t3:.2f}s encode+index {t5-t4:.2f}s",
      file=sys.stderr, flush=True)

qrows = sample_rows(n, B, 0)
Qh = dm.get_rows(qrows)
Q = torch.from_numpy(Qh).cuda()
oi = torch.empty((B, K), dtype=torch.int32, device="cuda"); od = torch.empty((B, K), dtype=torch.float32, device="cuda")
oc = torch.empty(B, dtype=torch.int32, device="cuda")


def step():
    N.check(L.gulon_grouped_index_batch_query_dev(index._h, Q.data_ptr(), B, K, 0, limit, oi.data_ptr(), od.data_ptr(),
                                                  oc.data_ptr(), None))


for _ in range(3):
    step()
torch.cuda.synchronize()
steps = 20
t = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / steps
ids = gv.perm[np.clip(oi.cpu().numpy(), 0, n - 1)]          # grouped position -> original row
rec, sd = recall_at_k(dm, Qh, K, ids.astype(np.int32), oc.cpu().numpy())
rows_per_query = float(np.mean(np.diff(np.r_[0, gv.offsets, n]))) * limit

from oracle import oracle                                    # CPU baseline leg only (the checker, timed)
codes = index.data.indices()
nq = 0
t = time.perf_counter()
while time.perf_counter() - t < 8.0 and nq < B:
    ei, ed, ec = oracle.grouped_query(codes, d, k, pq.flat_centroids(), gv.centroids, gv.offsets, Qh[nq:nq + 4], K, 0,
                                      limit)
    if nq == 0:
        first = (ei, ed, ec)
    nq += 4
cpu = (time.perf_counter() - t) / nq
oi_h, od_h = oi.cpu().numpy(), od.cpu().numpy()
same = bool(np.array_equal(first[0], oi_h[:4]) and np.array_equal(first[1].view(np.uint32), od_h[:4].view(np.uint32)))
print(json.dumps({"metric": "queries_per_sec", "value": B / dt, "unit": "queries/s", "ms_per_step": dt * 1e3,
                  "config": {"workload": f"GroupedIndex {n}x{d}, {len(gv.centroids)} groups, LimitGroups({limit}), "
                                         f"residual PQ(m={m},k={k}), batch={B}, K={K}",
                             "rows_scored_per_query": rows_per_query},
                  "recall_at_10": rec, "cpu_baseline": {"value": 1 / cpu, "unit": "queries/s", "cores": 1, "kind": "port",
                                                         "sample": f"first {nq} queries"},
                  "parity_vs_oracle": {"queries": 4, "ids_and_distances_equal": same},
                  "build_seconds": {"coarse_kmeans": t2 - t1, "group": t3 - t2, "residual_pq": t4 - t3,
                                    "encode": t5 - t4}}), flush=True)
