"""Tuning sweep of the quantized filter on the bench data: python scripts/filter_sweep.py [rows]"""
import ctypes as C
import itertools
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gulon_amd as g
from gulon_amd import native as N
from gulon_amd.recall import sample_rows

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d, m, k, B, K = 128, 16, 256, 1024, 10
L = N.lib()
dm = g.DeviceMatrix.synthetic(n, d, 3, 1234, 1000)
pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, 10))
enc = pq.encode(dm)
ix = g.PQIndex(pq, enc)
Qh = dm.get_rows(sample_rows(n, B, 0))
Q = torch.from_numpy(Qh).cuda()
oi = torch.empty((B, K), dtype=torch.int32, device="cuda"); od = torch.empty((B, K), dtype=torch.float32, device="cuda")
oc = torch.empty(B, dtype=torch.int32, device="cuda"); of = torch.empty(B, dtype=torch.int32, device="cuda")


def run(steps=5):
    for _ in range(2):
        N.check(L.gulon_index_batch_query_dev(ix._h, Q.data_ptr(), B, K, 0, n, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), of.data_ptr(), None))
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        N.check(L.gulon_index_batch_query_dev(ix._h, Q.data_ptr(), B, K, 0, n, oi.data_ptr(), od.data_ptr(), oc.data_ptr(), of.data_ptr(), None))
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps * 1e3


def tune(**kw):
    g.tune_live(**kw)


base = run()
ref_i, ref_d = oi.clone(), od.clone()
print(f"default: {base:.3f} ms", flush=True)
keys = ("GULON_FILTER_PERIOD", "GULON_FILTER_STAGE0", "GULON_FILTER_STAGE1", "GULON_FILTER_SAMPLE", "GULON_FILTER_NADD")
configs = [dict(zip(keys, c)) for c in itertools.product((128,), (0,), (3, 6, 12, 20), (16384, 32768, 65536), (2, 4))]
for cfg in configs:
    tune(**cfg)
    ms = run()
    same = bool(torch.equal(oi, ref_i) and torch.equal(od, ref_d))
    print(f"{cfg}: {ms:.3f} ms same={same}", flush=True)
