"""How much of the index could a query tile skip with reconstruction-space bounds?
Rows ordered by a coarse clustering of their PQ reconstructions x^; for a query q and a cluster (centre c, radius
R = max |x^ - c| over its rows) every row of the cluster has ADC distance |q - x^|^2 >= (|q - c| - R)^2.  A tile of 16
queries may skip the cluster if that bound exceeds the (K+1)-th ADC distance of ALL its queries.  Tiles of queries in
batch order against tiles of queries sorted by their nearest cluster.   python scripts/micro/block_skip_sim.py [rows] [clusters] [kind]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gulon_amd as g
from gulon_amd.recall import sample_rows

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 3
d, m, k, B, K = 128, 16, 256, 1024, 10
dev = torch.device("cuda", 0)
dm = g.DeviceMatrix.synthetic(n, d, kind, 1234, 1000)
pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, 10))
enc = pq.encode(dm)
codes = enc.indices()                                   # [m][n]
cents = pq.flat_centroids()
from gulon_amd.vectors import subvector_bounds
fr, un = subvector_bounds(d, m)
xh = torch.empty((n, d), dtype=torch.float32, device=dev)
for j in range(m):
    s = un[j] - fr[j]
    cj = torch.from_numpy(cents[k * fr[j]: k * fr[j] + k * s].reshape(k, s)).to(dev)
    xh[:, fr[j]:un[j]] = cj[torch.from_numpy(codes[j].astype(np.int64)).to(dev)]
Q = torch.from_numpy(dm.get_rows(sample_rows(n, B, 0))).to(dev)
# coarse k-means on the reconstructions
gen = torch.Generator(device="cpu").manual_seed(0)
C = xh[torch.randperm(n, generator=gen)[:nc].to(dev)].clone()
for it in range(8):
    a = torch.empty(n, dtype=torch.int64, device=dev)
    for i in range(0, n, 262144):
        x = xh[i:i + 262144]
        a[i:i + 262144] = ((x * x).sum(1, keepdim=True) - 2 * x @ C.T + (C * C).sum(1)[None, :]).argmin(1)
    cnt = torch.bincount(a, minlength=nc).clamp(min=1).float()
    C = torch.zeros_like(C).index_add_(0, a, xh) / cnt[:, None]
R = torch.zeros(nc, device=dev)
dist_to_c = (xh - C[a]).norm(dim=1)
R.scatter_reduce_(0, a, dist_to_c, reduce="amax")
size = torch.bincount(a, minlength=nc).float()
# true (K+1)-th ADC distance per query
tau = torch.empty(B, device=dev)
for q0 in range(0, B, 128):
    q = Q[q0:q0 + 128]
    best = torch.full((q.shape[0], K + 1), float("inf"), device=dev)
    for i in range(0, n, 262144):
        x = xh[i:i + 262144]
        dd = ((q * q).sum(1, keepdim=True) - 2 * q @ x.T + (x * x).sum(1)[None, :]).clamp(min=0)
        best = torch.cat([best, dd], 1).topk(K + 1, dim=1, largest=False).values
    tau[q0:q0 + 128] = best[:, K]
qc = torch.cdist(Q, C)                                   # |q - c|
lb = (qc - R[None, :]).clamp(min=0) ** 2                 # per (query, cluster)
need = lb <= tau[:, None] * 1.0001                       # cluster must be scanned for this query
w = size / size.sum()
print(f"rows {n}, {nc} clusters (sizes {int(size.min())}..{int(size.max())}), kind {kind}: a single query has to scan "
      f"{(need.float() * w[None, :]).sum(1).mean().item():.3f} of the rows on average")
for name, order in (("batch order", torch.arange(B, device=dev)), ("sorted by nearest cluster", qc.argmin(1).argsort())):
    nt = need[order].reshape(B // 16, 16, nc).any(1)     # tile needs the cluster if any of its 16 queries does
    print(f"  16-query tiles, {name}: {(nt.float() * w[None, :]).sum(1).mean().item():.3f} of the rows per tile")
# a smarter grouping: greedy tiles of queries with the most similar need sets is bounded below by the single-query figure
