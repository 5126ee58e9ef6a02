"""How often could a 64-row block stop early in the 8-bit filter?  Simulation on the bench data shape (numpy on the
host, codes / tables / bounds from the library): per 16-query tile and 64-row block, after j of the 16 quantizers,
do ALL 1024 (query, row) partial level sums already exceed QL?   python scripts/micro/early_exit_sim.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import gulon_amd as g
from gulon_amd.recall import sample_rows

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
d, m, k, K, B = 128, 16, 256, 10, 64
dm = g.DeviceMatrix.synthetic(n, d, 3, 1234, 1000)
pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, 10))
enc = pq.encode(dm)
codes = enc.indices()                      # [m][n]
ix = g.PQIndex(pq, enc)
Q = dm.get_rows(sample_rows(n, B, 0))
oi, od, oc, of = ix.batch_query_raw(K + 1, Q)
tau = od[:, K].astype(np.float64)          # the (K+1)-th distance: the best bound a filter stage can have
T = g.prepare_query(pq, Q).astype(np.float64)     # [B][m][k]
mins = T.min(axis=2)                       # [B][m]
for qmax, name in ((63, "6-bit levels (NADD=4)"), (31, "5-bit levels (NADD=8)"), (15, "4-bit levels, no widening")):
    QL = qmax - 1
    delta = (tau * (1 + 2 * m * 5.97e-8) - mins.sum(axis=1)) / QL
    lev = np.minimum(qmax, np.floor((T - mins[:, :, None]) / delta[:, None, None])).astype(np.int32)   # [B][m][k]
    nb = 2000                               # row blocks simulated
    rows = np.arange(nb * 64)
    out = {}
    surv = 0
    for t0 in range(0, B, 16):
        part = np.zeros((16, nb * 64), np.int32)
        alive_blocks = np.ones(nb, bool)
        for j in range(m):
            part += lev[t0:t0 + 16, j, :][:, codes[j, rows]]
            if j + 1 in (4, 6, 8, 10, 12):
                allout = (part > QL).reshape(16, nb, 64).all(axis=(0, 2))
                out.setdefault(j + 1, []).append(allout.mean())
        surv += (part <= QL).sum()
    print(name, "survivors per pair %.3g;" % (surv / (B * nb * 64)),
          "blocks that could stop after j quantizers:", {j: round(float(np.mean(v)), 3) for j, v in out.items()})
