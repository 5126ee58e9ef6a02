"""BASELINE config 1: 100k x 50 random vectors, exact L2 top-10 (Index.exactNearestNeighbours):
GPU kernel vs the CPU oracle on the same queries.  python tests/perf/bench_c1.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gulon_amd as g
from gulon_amd.recall import sample_rows
from oracle import oracle

n, d, B, K = 100_000, 50, 1024, 10
dm = g.DeviceMatrix.synthetic(n, d, 0, 1234, 1)
Q = dm.get_rows(sample_rows(n, B, 0))
g.exact_nearest_neighbours(dm, Q, K)                      # warm-up
t = time.perf_counter()
for _ in range(5):
    res = g.exact_nearest_neighbours(dm, Q, K)
gpu = (time.perf_counter() - t) / 5
X = dm.get_rows(np.arange(n, dtype=np.int32))
t = time.perf_counter()
nq = 64
oi, od, oc = oracle.exact_knn(X, Q[:nq], K)
cpu = (time.perf_counter() - t) / nq
ok = all(res[q].rows.tolist() == oi[q, :oc[q]].tolist() and
         np.array_equal(res[q].distances.view(np.uint32), od[q, :oc[q]].view(np.uint32)) for q in range(nq))
print(f"C1 exact kNN {n}x{d}, B={B}, K={K}: GPU {gpu * 1e3:.2f} ms per batch ({B / gpu:.0f} queries/s, host buffers in/out); "
      f"CPU oracle {cpu * 1e3:.2f} ms per query ({1 / cpu:.0f} queries/s, 1 core); first {nq} queries bit-exact: {ok}")
