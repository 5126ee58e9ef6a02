"""One-off stress run (not part of the suite) of ProductQuantizer.apply / encode against the CPU oracle: random sizes
(chunk boundaries of the streamed update: 8192 rows), dimensions, quantizers, cluster counts (the 256-, 512- and
1024-thread shapes of the chain kernel, and beyond 1024 where the regrouped update takes over), iteration counts, data
with duplicates / sparse rows / tiny and huge values.   python scripts/fuzz/fuzz_train.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gulon_amd as g
from oracle import oracle

oracle.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    n = int(rng.choice([rng.integers(300, 9000), rng.integers(8000, 70000), 8192 * int(rng.integers(1, 5)) + int(rng.integers(-2, 3))]))
    m = int(rng.integers(1, 9))
    s = int(rng.integers(1, 13))
    d = m * s + (int(rng.integers(0, m)) if rng.random() < 0.4 else 0)
    k = int(rng.choice([2, 7, 16, 100, 256, 256, 300, 512, 700, 1024, 1500]))
    k = min(k, n)
    iters = int(rng.integers(0, 6))
    kind = rng.choice(["normal", "dups", "sparse", "tiny", "huge", "uniform"])
    X = rng.standard_normal((n, d)).astype(np.float32)
    if kind == "dups":
        X[n // 2:] = X[: n - n // 2]
    elif kind == "sparse":
        X[rng.random((n, d)) < 0.8] = 0.0
    elif kind == "tiny":
        X[rng.random((n, d)) < 0.1] *= np.float32(1e-36)
    elif kind == "huge":
        X[rng.random((n, d)) < 0.02] *= np.float32(1e18)
    elif kind == "uniform":
        X = rng.random((n, d)).astype(np.float32)
    dm = g.DeviceMatrix.from_host(X)
    pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, iters))
    with np.errstate(all="ignore"):
        cents, _, _ = oracle.pq_train(X, m, k, iters)
    why = []
    got = pq.flat_centroids()
    if not np.array_equal(np.isnan(got), np.isnan(cents)) or not np.array_equal(got.view(np.uint32)[~np.isnan(cents)],
                                                                                cents.view(np.uint32)[~np.isnan(cents)]):
        why.append("centroids")
    else:
        enc = pq.encode(dm)
        if not np.array_equal(enc.indices(), oracle.pq_encode(X, m, k, cents)):
            why.append("codes")
    tag = f"case {case}: n={n} d={d} m={m} k={k} iterations={iters} data={kind}"
    if why:
        bad += 1
        print("MISMATCH", tag, why, flush=True)
    else:
        print("ok", tag, flush=True)
print(f"{cases - bad} of {cases} cases equal to the oracle")
sys.exit(1 if bad else 0)
