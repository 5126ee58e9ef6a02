"""One-off stress run (not part of the suite) of the filter's conflict-ordered code copy against the CPU oracle:
one-word indexes (m <= 16, k <= 256) of ragged sizes, ranges that cut blocks, duplicated rows (ties), K up to 63,
small filter ranges, the ordered and the plain copy of the same handle.   python scripts/fuzz/fuzz_order.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gulon_amd as g
from gulon_amd import native as N
from oracle import oracle

oracle.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
L = N.lib()
g.tune_live(GULON_FILTER_MIN_RB=4, GULON_FILTER_PERIOD=8, GULON_FILTER_STAGE0=1, GULON_FILTER_STAGE1=2, GULON_FILTER_SAMPLE=512)
bad = 0
for case in range(cases):
    m = int(rng.choice([16, 16, 16, 13, 9, 16]))
    s = int(rng.integers(1, 9))
    d = m * s + int(rng.integers(0, m)) * (rng.random() < 0.3)
    k = int(rng.choice([256, 256, 200, 64, 17]))
    n = int(rng.integers(3000, 90000))
    B = int(rng.integers(1, 70))
    K = int(rng.choice([1, 5, 10, 10, 33, 63]))
    frm = int(rng.integers(0, n // 2)) if rng.random() < 0.6 else 0
    until = int(rng.integers(frm + 1, n + 1)) if rng.random() < 0.6 else n
    cents = rng.standard_normal(k * d).astype(np.float32)
    idx = rng.integers(0, k, (m, n)).astype(np.int32)
    if rng.random() < 0.5:
        dup = int(rng.integers(1, n // 3))
        idx[:, -dup:] = idx[:, :dup]
    pq = g.ProductQuantizer.from_flat(k, d, m, cents)
    coder = pq.coder_factory(n)
    enc = g.EncodedMatrix(coder, [coder.build_code(idx[j]) for j in range(m)])
    Q = rng.standard_normal((B, d)).astype(np.float32)
    if rng.random() < 0.3:      # queries that are rows of the index
        from gulon_amd.vectors import subvector_bounds
        fr, un = subvector_bounds(d, m)
        for q in range(min(B, 8)):
            r = int(rng.integers(0, n))
            for j in range(m):
                w = un[j] - fr[j]
                c = idx[j, r]
                Q[q, fr[j]:un[j]] = cents[k * fr[j] + c * w: k * fr[j] + (c + 1) * w]
    ix = g.PQIndex(pq, enc)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K, frm, until)
    why = []
    for order in (1, 0):
        N.check(L.gulon_index_tuning(ix._h, b"GULON_FILTER_ORDER", order))
        res = ix.batch_query(K, Q, frm, until)
        for q, r in enumerate(res):
            if len(r) != oc[q] or not np.array_equal(r.distances.view(np.uint32), od[q, :oc[q]].view(np.uint32)):
                why.append(f"order={order} q={q}: distances")
                break
            if (r.flags == 0 or (r.flags & 4)) and r.rows.tolist() != oi[q, :oc[q]].tolist():
                why.append(f"order={order} q={q}: ids (flags {r.flags})")
                break
    ix.close()
    if why:
        bad += 1
        print("MISMATCH", dict(n=n, d=d, m=m, k=k, B=B, K=K, frm=frm, until=until), why, flush=True)
    elif case % 10 == 0:
        print("ok", case, dict(n=n, d=d, m=m, k=k, B=B, K=K, frm=frm, until=until), flush=True)
print("cases", cases, "mismatches", bad)
sys.exit(1 if bad else 0)
