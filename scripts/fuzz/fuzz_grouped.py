"""One-off stress run (not part of the suite) of the grouped index -- the by-group pre-selection with 8-bit bound tables
above all -- against the CPU oracle: random sizes, group counts, quantizers (m <= 16 and beyond), limits, k_nn (also
beyond 63), duplicated rows, tiny and far-away queries, queries that are rows.
    python scripts/fuzz/fuzz_grouped.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gulon_amd as g
from oracle import oracle

oracle.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    n = int(rng.integers(3000, 120000))
    m = int(rng.choice([16, 16, 16, 8, 4, 12, 20, 2]))
    s = int(rng.integers(1, 5))
    d = m * s + (int(rng.integers(0, m)) if rng.random() < 0.3 else 0)
    k = int(rng.choice([256, 256, 64, 16, 200]))
    groups = int(rng.integers(3, max(4, n // 150)))
    iters = int(rng.integers(1, 3))
    B = int(rng.integers(1, 48))
    K = int(rng.choice([1, 5, 10, 10, 10, 33, 63, 64, 150]))
    X = (rng.standard_normal((n, d)) * rng.choice([1.0, 0.01, 30.0]) + 3.0 * rng.integers(0, 4, (n, 1))).astype(np.float32)
    if rng.random() < 0.4:
        dup = int(rng.integers(1, n // 3))
        X[-dup:] = X[:dup]
    dm = g.DeviceMatrix.from_host(X)
    coarse = g.KMeans.compute_clusters(g.Vectors(dm), g.KMeansConfig(groups, iters))
    gv = g.group(dm, coarse)
    pq = g.ProductQuantizer.apply(gv.residuals, g.ProductQuantizerConfig(k, m, iters))
    ng = len(gv.centroids)
    if rng.random() < 0.8:
        strategy, limit = 0, int(rng.integers(1, ng + 3))
        strat = g.LimitGroups(limit)
    else:
        strategy, limit = 1, int(rng.integers(1, 2 * n))
        strat = g.LimitVectors(limit)
    index = g.Index.grouped(gv, pq, strat)
    Q = X[rng.integers(0, n, B)].copy()
    Q[: B // 3] += (rng.standard_normal((B // 3, d)) * 0.05).astype(np.float32)
    if B > 4 and rng.random() < 0.3:
        Q[1] *= np.float32(1e3)
        Q[2] *= np.float32(1e-6)
    oi, od, oc = index.batch_query_raw(K, Q)
    ei, ed, ec = oracle.grouped_query(index.data.indices(), d, k, pq.flat_centroids(), gv.centroids, gv.offsets, Q, K, strategy,
                                      limit)
    why = []
    if not np.array_equal(oc, ec):
        why.append("counts")
    else:
        for q in range(B):
            if oi[q, :oc[q]].tolist() != ei[q, :ec[q]].tolist():
                why.append(f"q={q}: ids")
                break
            if not np.array_equal(od[q, :oc[q]].view(np.uint32), ed[q, :ec[q]].view(np.uint32)):
                why.append(f"q={q}: distances")
                break
    index.close()
    tag = f"case {case}: n={n} d={d} m={m} k={k} groups={ng} strategy={strategy} limit={limit} B={B} K={K}"
    if why:
        bad += 1
        print("MISMATCH", tag, why, flush=True)
    else:
        print("ok", tag, flush=True)
print(f"{cases - bad} of {cases} cases equal to the oracle")
sys.exit(1 if bad else 0)
