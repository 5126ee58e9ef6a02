"""One-off stress run (not part of the suite): random shapes through the sharded pipeline with shared
bounds (threads as ranks, tests/test_gpu_shared_bounds.ThreadGroup) against the unsharded index and the
CPU oracle.   python scripts/fuzz/fuzz_sharded.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.cuda.init()
import gulon_amd as g
from gulon_amd import native as N
from oracle import oracle
from test_gpu_shared_bounds import sharded_query

oracle.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
L = N.lib()


def tune(**kw):
    g.tune_live(**kw)


bad = 0
for case in range(cases):
    d = int(rng.integers(8, 97))
    m = int(rng.integers(2, min(d, 36) + 1))
    k = int(rng.choice([2, 16, 100, 256]))
    n = int(rng.integers(3000, 120000))
    B = int(rng.integers(1, 70))
    K = int(rng.choice([1, 5, 10, 31, 63]))
    world = int(rng.integers(2, 6))
    dup = int(rng.integers(0, n // 5)) if case % 3 == 0 else 0
    tune(GULON_FILTER_MIN_RB=int(rng.choice([4, 64, 512])), GULON_FILTER_PERIOD=int(rng.choice([4, 8, 128])),
         GULON_FILTER_STAGE1=int(rng.choice([1, 2, 6])), GULON_FILTER_SAMPLE=int(rng.choice([512, 4096, 65536])),
         GULON_FILTER_SHARED_STAGE1=int(rng.choice([-1, 0, 1])), GULON_FILTER_NADD=int(rng.choice([0, 2, 4])))
    r2 = np.random.default_rng(1000 + case)
    cents = r2.standard_normal(k * d).astype(np.float32)
    if case % 2 == 0:      # clustered codes: rows near a few prototypes, so that bounds separate
        proto = r2.integers(0, k, (m, 50))
        pick = r2.integers(0, 50, n)
        idx = proto[:, pick].astype(np.int32)
        flip = r2.random((m, n)) < 0.2
        idx = np.where(flip, r2.integers(0, k, (m, n)), idx).astype(np.int32)
    else:
        idx = r2.integers(0, k, (m, n)).astype(np.int32)
    if dup:
        idx[:, -dup:] = idx[:, :dup]
    pq = g.ProductQuantizer.from_flat(k, d, m, cents)
    coder = pq.coder_factory(n)
    enc = g.EncodedMatrix(coder, [coder.build_code(idx[j]) for j in range(m)])
    Q = r2.standard_normal((B, d)).astype(np.float32)
    full = g.PQIndex(pq, enc).batch_query_raw(K, Q)
    sh = sharded_query(g, pq, enc, n, world, Q, K)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K)
    # distances and counts: always the oracle's, bit for bit.  Ids: wherever a result carries no tie flag or was
    # replayed exactly.  (The sharded pipeline replays at most GULON_REPLAY_MAX_FLAGGED = 16 flagged queries per
    # batch from pools of 2048 candidates, the unsharded one more: the replay bit may differ, the tie bits may not.)
    why = []
    for name, r in (("unsharded", full), ("sharded", sh)):
        if not np.array_equal(od.view(np.uint32), r[1].view(np.uint32)):
            why.append(name + ":distances")
        if not np.array_equal(oc, r[2]):
            why.append(name + ":counts")
        for q in range(B):
            if (r[3][q] == 0 or (r[3][q] & 4)) and not np.array_equal(oi[q, :oc[q]], r[0][q, :oc[q]]):
                why.append(f"{name}:ids q={q} flags={r[3][q]}")
                break
    if not np.array_equal(full[3] & 3, sh[3] & 3):
        why.append("tie flags differ")
    ok = not why
    print(f"case {case}: n={n} d={d} m={m} k={k} B={B} K={K} world={world} dup={dup} -> {'ok' if ok else 'MISMATCH ' + '; '.join(why)}",
          flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
