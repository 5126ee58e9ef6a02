"""Wide-code benchmark: PQ with more than 256 centroids per quantizer (Coder.BytePlus
widths 10/12/16) -> flat index -> batched queries.   python tests/perf/bench_wide.py [rows] [clusters] [dim] [m]"""
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
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
d = int(sys.argv[3]) if len(sys.argv) > 3 else 128
m = int(sys.argv[4]) if len(sys.argv) > 4 else 16
B, K, iters = 1024, 10, 10
L = N.lib()
t0 = time.perf_counter()
dm = g.DeviceMatrix.synthetic(n, d, 3, 1234, 1000)
t1 = time.perf_counter()
pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, iters))
t2 = time.perf_counter()
enc = pq.encode(dm)
t3 = time.perf_counter()
index = g.PQIndex(pq, enc)
print(f"[wide] n={n} d={d} m={m} k={k} (code width {enc.coder.width}): synth {t1-t0:.2f}s train {t2-t1:.2f}s "
      f"encode {t3-t2:.2f}s", file=sys.stderr, flush=True)
Qh = dm.get_rows(sample_rows(n, B, 0))
Q = torch.from_numpy(Qh).cuda()
oi = torch.empty((B, K), dtype=torch.int32, device="cuda"); od = torch.empty((B, K), dtype=torch.float32, device="cuda")
oc = torch.empty(B, dtype=torch.int32, device="cuda"); of = torch.empty(B, dtype=torch.int32, device="cuda")


def step():
    N.check(L.gulon_index_batch_query_dev(index._h, Q.data_ptr(), B, K, 0, n, oi.data_ptr(), od.data_ptr(), oc.data_ptr(),
                                          of.data_ptr(), None))


for _ in range(2):
    step()
torch.cuda.synchronize()
steps = 10
t = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / steps
mean, sd = recall_at_k(dm, Qh, K, oi.cpu().numpy(), oc.cpu().numpy())
res = {"metric": "queries_per_sec", "value": B / dt, "unit": "queries/s", "ms_per_step": dt * 1e3,
       "config": {"workload": f"flat PQ index {n}x{d}, m={m}, k={k} (code width {enc.coder.width}), batch={B}, K={K}"},
       "recall_at_10": mean, "build_seconds": {"train": t2 - t1, "encode": t3 - t2},
       "algorithmic_lookups_per_s": B * n * m / dt}
from oracle import oracle                                    # checker: a bounded sample on the host
nq = 4
t = time.perf_counter()
ei, ed, ec = oracle.pq_batch_query(enc.indices(), d, k, pq.flat_centroids(), Qh[:nq], K)
cpu = time.perf_counter() - t
flg = of.cpu().numpy()[:nq]
same_d = bool(np.array_equal(ed.view(np.uint32), od.cpu().numpy()[:nq].view(np.uint32)))
same_i = bool(all(flg[q] != 0 or np.array_equal(ei[q], oi.cpu().numpy()[q]) for q in range(nq)))
res["cpu_baseline"] = {"value": nq / cpu, "unit": "queries/s", "cores": 1, "kind": "port", "sample": f"first {nq} queries"}
res["parity_vs_oracle"] = {"queries": nq, "distances_bit_exact": same_d, "ids_equal": same_i}
print(json.dumps(res))
