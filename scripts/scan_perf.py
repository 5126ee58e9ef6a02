"""Quick timing of the ADC scan on random codes (no training): python scripts/scan_perf.py [n] [m] [B] [K]"""
import ctypes as C
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gulon_amd as g
from gulon_amd import native as N

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 16
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
K = int(sys.argv[4]) if len(sys.argv) > 4 else 10
d = m * 8
k = 256
rng = np.random.default_rng(0)
cents = rng.standard_normal(k * d).astype(np.float32)
codes = rng.integers(0, 256, (m, n), dtype=np.uint8)
pq = g.ProductQuantizer.from_flat(k, d, m, cents)
coder = pq.coder_factory(n)
enc = g.EncodedMatrix(coder, [codes[j] for j in range(m)])
t0 = time.perf_counter()
ix = g.PQIndex(pq, enc)
print("index_create %.2fs" % (time.perf_counter() - t0), flush=True)
Q = rng.standard_normal((B, d)).astype(np.float32)
L = N.lib()
dq, oi, od, oc, of = (C.c_void_p() for _ in range(5))
N.check(L.gulon_dev_malloc(C.byref(dq), Q.nbytes))
N.check(L.gulon_dev_malloc(C.byref(oi), B * K * 4))
N.check(L.gulon_dev_malloc(C.byref(od), B * K * 4))
N.check(L.gulon_dev_malloc(C.byref(oc), B * 4))
N.check(L.gulon_dev_malloc(C.byref(of), B * 4))
N.check(L.gulon_memcpy_h2d(dq, Q.ctypes.data_as(C.c_void_p), Q.nbytes))
for it in range(2):
    N.check(L.gulon_index_batch_query_dev(ix._h, dq, B, K, 0, n, oi, od, oc, of, None))
N.check(L.gulon_device_synchronize())
steps = 5
t0 = time.perf_counter()
for it in range(steps):
    N.check(L.gulon_index_batch_query_dev(ix._h, dq, B, K, 0, n, oi, od, oc, of, None))
N.check(L.gulon_device_synchronize())
dt = (time.perf_counter() - t0) / steps
alg = B * n * m
print(f"n={n} m={m} B={B} K={K}: {dt*1e3:.3f} ms/batch  {B/dt:.0f} qps  alg {alg/dt/1e12:.3f} TB/s = {alg/dt/8e12*100:.1f}% of 8 TB/s", flush=True)
