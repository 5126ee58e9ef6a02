"""One-off stress run (not part of the suite) for round 2's new query paths, against the CPU oracle:
  * the quantized filter over 16-bit code words (wide_filter.hip): random k in 257..2048, m, ranges, K, ties;
  * the tie replay's long level through the quantized filter (replay_level2_filtered): many flagged queries over
    ranges long enough for a third level, every batch queried twice (the handle switches roads after the first).
python scripts/fuzz/fuzz_round2.py [cases] [seed]"""
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
bad = 0


def check(name, res, oi, od, oc, need_replay=False):
    why = []
    if not np.array_equal(od.view(np.uint32), res[1].view(np.uint32)):
        why.append(name + ":distances")
    if not np.array_equal(oc, res[2]):
        why.append(name + ":counts")
    for q in range(len(oc)):
        fl = res[3][q]
        if (fl == 0 or (fl & 4)) and not np.array_equal(oi[q, :oc[q]], res[0][q, :oc[q]]):
            why.append(f"{name}:ids q={q} flags={fl}")
            break
        if need_replay and (fl & 3) and not (fl & 4):
            why.append(f"{name}:not replayed q={q}")
            break
    return why


for case in range(cases):
    r2 = np.random.default_rng(5000 + case)
    wide = case % 2 == 0
    if wide:
        k = int(rng.choice([257, 300, 512, 777, 1024, 2048]))
        m = int(rng.integers(2, 17))
        if m * k * 4 > 128 * 1024:
            m = max(2, 128 * 1024 // (4 * k))
        d = m * int(rng.integers(1, 5))
        n = int(rng.integers(20000, 200000))
        B = int(rng.integers(1, 60))
        K = int(rng.choice([1, 5, 10, 31, 63]))
        dup = int(rng.integers(0, n // 4)) if case % 4 == 0 else 0
        nbase = 0
    else:
        k, m = 256, int(rng.choice([4, 8, 16, 12]))
        d = m * int(rng.integers(1, 5))
        n = int(rng.integers(180000, 420000))
        B = int(rng.integers(40, 120))
        K = int(rng.choice([1, 5, 10, 31]))
        dup, nbase = 0, int(rng.integers(50, 600))
    frm = int(rng.integers(0, n // 10)) if case % 3 == 0 else 0
    until = n - int(rng.integers(0, n // 10)) if case % 3 == 0 else n
    cents = r2.standard_normal(k * d).astype(np.float32)
    idx = r2.integers(0, k, (m, n)).astype(np.int32)
    if dup:
        idx[:, -dup:] = idx[:, :dup]
    if nbase:
        base = r2.integers(0, k, (m, nbase)).astype(np.int32)
        copies = r2.permutation(n)[:n // 2]
        idx[:, copies] = base[:, r2.integers(0, nbase, n // 2)]
    pq = g.ProductQuantizer.from_flat(k, d, m, cents)
    coder = pq.coder_factory(n)
    enc = g.EncodedMatrix(coder, [coder.build_code(idx[j]) for j in range(m)])
    ix = g.PQIndex(pq, enc)
    N.check(L.gulon_index_tuning(ix._h, b"GULON_FILTER_MIN_RB", int(rng.choice([4, 64, 512]))))
    if nbase:
        rows = copies[:B]
    else:
        rows = r2.integers(0, n, B)
    Q = np.stack([ix.decode(int(r)) for r in rows]).astype(np.float32)
    if case % 5 == 0:
        Q[0] = r2.standard_normal(d)
    oi, od, oc = oracle.pq_batch_query(idx, d, k, cents, Q, K, frm, until)
    why = []
    for rep in range(2):
        why += check(f"pass{rep}", ix.batch_query_raw(K, Q, frm, until), oi, od, oc, need_replay=bool(nbase))
    ix.close()
    ok = not why
    print(f"case {case}: {'wide' if wide else 'ties'} n={n} d={d} m={m} k={k} B={B} K={K} [{frm},{until}) dup={dup} nbase={nbase} -> "
          f"{'ok' if ok else 'MISMATCH ' + '; '.join(why)}", flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
