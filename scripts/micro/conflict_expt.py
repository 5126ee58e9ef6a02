import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gulon_amd as g
from gulon_amd import native as N
n, m, B, K = 10_000_000, 16, 1024, 10
d, k = m * 8, 256
rng = np.random.default_rng(0)
cents = rng.standard_normal(k * d).astype(np.float32)
L = N.lib()
Q = rng.standard_normal((B, d)).astype(np.float32)
rows = np.arange(n, dtype=np.int64)
for mode in ("random", "low4=lane", "low3=lane"):
    codes = rng.integers(0, 256, (m, n), dtype=np.uint8)
    if mode == "low4=lane":
        codes = (codes & 0xF0) | (rows & 15).astype(np.uint8)
    elif mode == "low3=lane":
        codes = (codes & 0xF8) | (rows & 7).astype(np.uint8)
    pq = g.ProductQuantizer.from_flat(k, d, m, cents)
    coder = pq.coder_factory(n)
    ix = g.PQIndex(pq, g.EncodedMatrix(coder, [codes[j] for j in range(m)]))
    dq, oi, od, oc, of = (C.c_void_p() for _ in range(5))
    for p, sz in ((dq, Q.nbytes), (oi, B * K * 4), (od, B * K * 4), (oc, B * 4), (of, B * 4)):
        N.check(L.gulon_dev_malloc(C.byref(p), sz))
    N.check(L.gulon_memcpy_h2d(dq, Q.ctypes.data_as(C.c_void_p), Q.nbytes))
    for it in range(2):
        N.check(L.gulon_index_batch_query_dev(ix._h, dq, B, K, 0, n, oi, od, oc, of, None))
    N.check(L.gulon_device_synchronize())
    t0 = time.perf_counter()
    for it in range(5):
        N.check(L.gulon_index_batch_query_dev(ix._h, dq, B, K, 0, n, oi, od, oc, of, None))
    N.check(L.gulon_device_synchronize())
    print(mode, f"{(time.perf_counter() - t0) / 5 * 1e3:.3f} ms/batch", flush=True)
    ix.close()
