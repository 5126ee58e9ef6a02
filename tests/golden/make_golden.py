#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the CPU oracle.

SELF-GENERATED: these vectors were produced by oracle/gulon_oracle.c (a restatement of the
Scala reference), NOT by the JVM reference -- the reference ships no golden vectors and
cannot be run in this image.  They pin the oracle against silent drift and give the GPU
tests committed inputs/outputs.  If a JVM ever becomes available, dump the same arrays
from the real reference and diff.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as o  # noqa: E402


def main():
    # 1. java.util.Random streams (JDK algorithm)
    r0, r42 = o.JavaRandom(0), o.JavaRandom(42)
    rb = o.JavaRandom(0)
    np.savez(os.path.join(HERE, "java_random.npz"),
             seed0_next_int=np.array([r0.next_int() for _ in range(16)], np.int64),
             seed42_next_int_10=np.array([r42.next_int(10) for _ in range(16)], np.int64),
             seed0_next_int_10m=np.array([o.JavaRandom(0).next_int(10_000_000)], np.int64),
             seed0_booleans=np.array([rb.next_boolean() for _ in range(64)], np.uint8))

    # 2. Vectors.subvectors splits
    sv = {}
    for d, m in [(50, 25), (128, 16), (300, 32), (1024, 64)]:
        fr, un = o.subvectors(d, m)
        sv[f"from_{d}_{m}"], sv[f"until_{d}_{m}"] = fr, un
    np.savez(os.path.join(HERE, "subvectors.npz"), **sv)

    # 3. small PQ end to end: 2000 x 16, m = 4, k = 16
    n, d, m, k, iters = 2000, 16, 4, 16, 10
    X = o.synth(n, d, 1, 7, 12)
    fr, un = o.subvectors(d, m)
    C0, rows0 = o.kmeans_init(X, int(fr[1]), int(un[1] - fr[1]), k, 1)
    a0 = o.kmeans_assign(X, int(fr[1]), int(un[1] - fr[1]), C0, 25000)
    C1 = o.kmeans_from_assignment(X, int(fr[1]), int(un[1] - fr[1]), k, a0)
    cents, its, conv = o.pq_train(X, m, k, iters)
    idx = o.pq_encode(X, m, k, cents)
    Q = X[[3, 500, 1234, 1999]]
    T = o.prepare_query(cents, d, m, k, Q)
    oi, od, oc = o.pq_batch_query(idx, d, k, cents, Q, 10)
    ei, ed, ec = o.exact_knn(X, Q, 10)
    np.savez_compressed(os.path.join(HERE, "pq_small.npz"), X=X, init_rows_q1=rows0, init_centroids_q1=C0,
                        assign0_q1=a0, centroids1_q1=C1, codebooks=cents, iterations=its, converged=conv,
                        codes=idx.astype(np.uint8), queries=Q, tables=T, nn_idx=oi, nn_dist=od, nn_count=oc,
                        exact_idx=ei, exact_dist=ed)

    # 4. tie-break stream across the 25 000-row parAssign boundary (duplicate / zero centroids)
    rng = np.random.default_rng(5)
    Xt = rng.standard_normal((60000, 2)).astype(np.float32)
    Cz = np.zeros((4, 2), np.float32)
    Cd = np.array([[0.5, 0.5], [0.5, 0.5], [-1, 0], [-1, 0], [0.5, 0.5]], np.float32)
    np.savez_compressed(os.path.join(HERE, "tie_break.npz"), X=Xt, C_zero=Cz, C_dup=Cd,
                        serial_zero=o.kmeans_assign(Xt, 0, 2, Cz, 0).astype(np.uint8),
                        par_zero=o.kmeans_assign(Xt, 0, 2, Cz, 25000).astype(np.uint8),
                        serial_dup=o.kmeans_assign(Xt, 0, 2, Cd, 0).astype(np.uint8),
                        par_dup=o.kmeans_assign(Xt, 0, 2, Cd, 25000).astype(np.uint8))
    # 5. GroupedIndex end to end on the pq_small data: 12 coarse clusters, residual PQ m = 4, k = 16
    gcfg = dict(groups=12, m=4, k=16, iters=5)
    Cc, _ = o.kmeans_compute_clusters(X, 0, d, gcfg["groups"], gcfg["iters"])
    ga = o.kmeans_assign(X, 0, d, Cc, 25000)
    perm, gcent, goff = o.group_rows(ga, Cc)
    R = o.group_residuals(X, perm, gcent, goff)
    rc, _, _ = o.pq_train(R, gcfg["m"], gcfg["k"], gcfg["iters"])
    ridx = o.pq_encode(R, gcfg["m"], gcfg["k"], rc)
    Qg = X[[7, 400, 1500, 1999, 3]]
    out = {}
    for name, (strat, lim) in {"groups3": (0, 3), "vectors600": (1, 600)}.items():
        gi, gd, gc = o.grouped_query(ridx, d, gcfg["k"], rc, gcent, goff, Qg, 10, strat, lim)
        out[f"{name}_idx"], out[f"{name}_dist"], out[f"{name}_count"] = gi, gd, gc
    np.savez_compressed(os.path.join(HERE, "grouped_small.npz"), coarse_centroids=Cc, assignments=ga.astype(np.uint8),
                        perm=perm, group_centroids=gcent, offsets=goff, residual_codebooks=rc,
                        residual_codes=ridx.astype(np.uint8), queries=Qg, **out)
    # 6. more than 256 centroids per quantizer (Coder.BytePlus, width 10): 3000 x 12, m = 3, k = 300
    wn, wd, wm, wk = 3000, 12, 3, 300
    Xw = o.synth(wn, wd, 3, 11, 40)
    wc, wits, wconv = o.pq_train(Xw, wm, wk, 3)
    widx = o.pq_encode(Xw, wm, wk, wc)
    Qw = Xw[[0, 17, 1500, 2999]]
    wi, wdist, wcnt = o.pq_batch_query(widx, wd, wk, wc, Qw, 5)
    np.savez_compressed(os.path.join(HERE, "pq_wide.npz"), X=Xw, codebooks=wc, iterations=wits, converged=wconv,
                        codes=widx.astype(np.uint16), packed_q0=o.coder_build(10, widx[0]), queries=Qw,
                        tables=o.prepare_query(wc, wd, wm, wk, Qw), nn_idx=wi, nn_dist=wdist, nn_count=wcnt)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
