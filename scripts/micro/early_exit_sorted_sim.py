"""Early exit of the 8-bit filter when the rows of a 64-row block are NEIGHBOURS: rows ordered by a coarse k-means
cluster of the full vectors (what GroupedIndex does to its rows), or lexicographically by their first codes, against
the index's own order.  Per 16-query tile and 64-row block, after j of the 16 quantizers: do ALL 1024 (query, row)
partial level sums already exceed the budget?  Same simulation as early_exit_sim.py (numpy on the host; codes, tables
and bounds from the library).     python scripts/micro/early_exit_sorted_sim.py [rows] [coarse clusters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import gulon_amd as g
from gulon_amd.recall import sample_rows

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
kc = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 3
d, m, k, K, B = 128, 16, 256, 10, 64
dm = g.DeviceMatrix.synthetic(n, d, kind, 1234, 1000)
pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, 10))
enc = pq.encode(dm)
codes = enc.indices()                      # [m][n]
ix = g.PQIndex(pq, enc)
Q = dm.get_rows(sample_rows(n, B, 0))
oi, od, oc, of = ix.batch_query_raw(K + 1, Q)
tau = od[:, K].astype(np.float64)          # the (K+1)-th distance: the best bound a filter stage can have
T = g.prepare_query(pq, Q).astype(np.float64)     # [B][m][k]
mins = T.min(axis=2)
from gulon_amd.kmeans import KMeans as _KM, Config as _KC
coarse = _KM.compute_clusters(g.Vectors(dm), _KC(kc, 3, 7))
assign = coarse.par_assign(g.Vectors(dm))
orders = {
    "index order": np.arange(n),
    "by coarse cluster (k=%d)" % kc: np.argsort(assign, kind="stable"),
    "by (code0, code1, code2)": np.lexsort((codes[2], codes[1], codes[0])),
}
qmax = 63
QL = qmax - 1
for tau_scale, tname in ((1.0, "true (K+1)-th distance"), (1.14, "1.14 x (after the first stage)")):
    delta = (tau * tau_scale * (1 + 2 * m * 5.97e-8) - mins.sum(axis=1)) / QL
    lev = np.minimum(qmax, np.floor((T - mins[:, :, None]) / delta[:, None, None])).astype(np.int32)   # [B][m][k]
    for oname, order in orders.items():
        nb = 4000
        rng = np.random.default_rng(1)
        blocks = rng.choice(n // 64, nb, replace=False)          # random blocks of the ordered rows
        rows = order[(blocks[:, None] * 64 + np.arange(64)[None, :]).reshape(-1)]
        out, work = {}, []
        for t0 in range(0, B, 16):
            part = np.zeros((16, nb * 64), np.int32)
            alive = np.ones(nb, bool)
            done_at = np.full(nb, m)
            for j in range(m):
                part += lev[t0:t0 + 16, j, :][:, codes[j, rows]]
                allout = (part > QL).reshape(16, nb, 64).all(axis=(0, 2))
                newly = allout & alive
                done_at[newly] = j + 1
                alive &= ~allout
                if j + 1 in (2, 4, 6, 8, 10, 12):
                    out.setdefault(j + 1, []).append((~alive).mean())
            work.append(done_at.mean())
        print(f"bound = {tname}; rows {oname}: blocks stopped after j quantizers",
              {j: round(float(np.mean(v)), 3) for j, v in out.items()}, "mean look-ups per block %.2f of 16" % np.mean(work), flush=True)
