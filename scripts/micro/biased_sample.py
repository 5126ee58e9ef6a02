"""Experiment: how tight is the pruning bound from a BIASED sample (rows of the (code_0, code_1) cells with the
smallest T_0 + T_1) against the uniform sample + first stage the filter uses today?  python scripts/micro/biased_sample.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gulon_amd as g
from gulon_amd.recall import sample_rows

n, d, m, k, B, K = 10_000_000, 128, 16, 256, 32, 10
dm = g.DeviceMatrix.synthetic(n, d, 3, 1234, 1000)
pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, 10))
enc = pq.encode(dm)
codes = torch.from_numpy(np.stack(enc.encodings)).cuda().long()        # [m][n]
Qh = dm.get_rows(sample_rows(n, 1024, 0))[:B]
T = torch.from_numpy(g.prepare_query(pq, Qh)).cuda()                   # [B][m][k]
res = {"true": [], "uniform523k": [], "uniform54k": []}
for R in (4096, 16384, 65536):
    for nq in (1, 2, 3):
        res[f"cells{nq}_R{R}"] = []
gen = torch.Generator(device="cuda").manual_seed(1)
for q in range(B):
    dist = torch.zeros(n, device="cuda")
    for j in range(m):
        dist += T[q, j][codes[j]]
    true = torch.kthvalue(dist, K + 1).values.item()
    res["true"].append(true)
    for name, s in (("uniform523k", 523000), ("uniform54k", 54000)):
        idx = torch.randint(0, n, (s,), device="cuda", generator=gen)
        res[name].append(torch.kthvalue(dist[idx], K + 1).values.item())
    for nq in (1, 2, 3):
        score = torch.zeros(n, device="cuda")
        for j in range(nq):
            score += T[q, j][codes[j]]
        for R in (4096, 16384, 65536):
            idx = torch.topk(score, R, largest=False).indices
            res[f"cells{nq}_R{R}"].append(torch.kthvalue(dist[idx], K + 1).values.item())
true = np.array(res["true"])
for name, v in res.items():
    v = np.array(v)
    frac = [(float((v[i] >= 0))) for i in range(B)]
    print(f"{name:18s} bound / true (K+1)-th distance: mean {np.mean(v / true):.4f}  max {np.max(v / true):.4f}")
