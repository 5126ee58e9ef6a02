"""Experiment: bound quality of the side x side grid of best (code_0, code_1) cells, per query (worst cases)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gulon_amd as g
from gulon_amd.recall import sample_rows

n, d, m, k, B, K = 10_000_000, 128, 16, 256, 256, 10
dm = g.DeviceMatrix.synthetic(n, d, 3, 1234, 1000)
pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, 10))
enc = pq.encode(dm)
codes = torch.from_numpy(np.stack(enc.encodings)).cuda().long()        # [m][n]
Qh = dm.get_rows(sample_rows(n, 1024, 0))[:B]
T = torch.from_numpy(g.prepare_query(pq, Qh)).cuda()                   # [B][m][k]
key = codes[0] * 256 + codes[1]
order = torch.argsort(key, stable=True)
counts = torch.bincount(key, minlength=65536)
off = torch.cat([torch.zeros(1, dtype=torch.long, device="cuda"), torch.cumsum(counts, 0)])
out = []
for q in range(B):
    dist = torch.zeros(n, device="cuda")
    for j in range(m):
        dist += T[q, j][codes[j]]
    true = torch.kthvalue(dist, K + 1).values.item()
    row = {"q": q, "true": true}
    for side, R in ((16, 8192), (16, 32768), (8, 8192)):
        a = torch.topk(T[q, 0], side, largest=False)
        b = torch.topk(T[q, 1], side, largest=False)
        score = (a.values[:, None] + b.values[None, :]).reshape(-1)
        cell = (a.indices[:, None] * 256 + b.indices[None, :]).reshape(-1)
        o = torch.argsort(score)
        rows, tot = [], 0
        for c in cell[o].tolist():
            lo, hi = off[c].item(), off[c + 1].item()
            if hi > lo:
                rows.append(order[lo:hi]); tot += hi - lo
            if tot >= R:
                break
        rows = torch.cat(rows)[:R] if rows else torch.zeros(0, dtype=torch.long, device="cuda")
        row[f"s{side}_R{R}"] = (torch.kthvalue(dist[rows], K + 1).values.item() / true) if len(rows) > K else float("inf")
        row[f"n{side}_R{R}"] = int(len(rows))
    out.append(row)
for name in ("s16_R8192", "s16_R32768", "s8_R8192"):
    v = np.array([r[name] for r in out])
    nn = np.array([r["n" + name[1:]] for r in out])
    w = np.argsort(-v)[:5]
    print(name, "mean %.4f p90 %.4f p99 %.4f max %.4f" % (v.mean(), np.quantile(v, .9), np.quantile(v, .99), v.max()),
          "| worst:", [(int(i), round(float(v[i]), 3), int(nn[i])) for i in w])
