"""Filtered vs exact scan as a function of the row count (bench data): python scripts/filter_crossover.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gulon_amd as g
from gulon_amd import native as N
from gulon_amd.recall import sample_rows

d, m, k, B, K = 128, 16, 256, 1024, 10
L = N.lib()
oi = torch.empty((B, K), dtype=torch.int32, device="cuda"); od = torch.empty((B, K), dtype=torch.float32, device="cuda")
oc = torch.empty(B, dtype=torch.int32, device="cuda"); of = torch.empty(B, dtype=torch.int32, device="cuda")
for n in (65536, 131072, 262144, 524288, 1250000, 2500000):
    dm = g.DeviceMatrix.synthetic(n, d, 3, 1234, 1000)
    pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, 5))
    ix = g.PQIndex(pq, pq.encode(dm))
    Q = torch.from_numpy(dm.get_rows(sample_rows(n, B, 0))).cuda()
    res = {}
    for name, min_rb in (("filter", 4), ("exact", 1 << 30)):
        g.tune_live(GULON_FILTER_MIN_RB=min_rb)
        for _ in range(3):
            N.check(L.gulon_index_batch_query_dev(ix._h, Q.data_ptr(), B, K, 0, n, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), of.data_ptr(), None))
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            N.check(L.gulon_index_batch_query_dev(ix._h, Q.data_ptr(), B, K, 0, n, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), of.data_ptr(), None))
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t) / 10 * 1e3
        res[name + "_ids"] = oi.clone()
    same = bool(torch.equal(res["filter_ids"], res["exact_ids"]))
    print(f"n={n:8d}: filter {res['filter']:.3f} ms   exact {res['exact']:.3f} ms   same={same}", flush=True)
    ix.close()
