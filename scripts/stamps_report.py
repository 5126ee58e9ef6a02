"""Where a main-stage filter launch spends its time, from the per-wave stamps of an experiment build
(scripts/variant.sh stamps filter -DGULON_FILTER_STAMPS; GULON_FILTER_STAMPS=<file>):
python scripts/stamps_report.py <file> [cus to list]"""
import sys

import numpy as np

raw = open(sys.argv[1], "rb").read()
ftiles, nchunks, e_count, per = np.frombuffer(raw[:16], np.int32)
st = np.frombuffer(raw[16:], np.uint64).reshape(nchunks * ftiles, 52)
w = st[:, :48].reshape(-1, 16, 3).astype(np.int64)
base = w[:, :, 0].min()
us = lambda x: (x - base) / 100.0          # 100 MHz wall clock
start, staged, end = us(w[:, :, 0]), us(w[:, :, 1]), us(w[:, :, 2])
hw, xcc = (st[:, 48] & 0xFFFFFFFF).astype(np.int64), (st[:, 48] >> 32).astype(np.int64) & 0xF
print(f"grid {ftiles} tiles x {nchunks} chunks = {ftiles * nchunks} workgroups, {e_count} row blocks, {per} per chunk")
print(f"kernel span (first wave start .. last wave end): {end.max():.1f} us")
wg_start, wg_end = start.min(1), end.max(1)
print(f"wave start skew inside a workgroup (us): median {np.median(start.max(1) - start.min(1)):.1f} max {(start.max(1) - start.min(1)).max():.1f}")
print(f"table staging (us): median {np.median(staged - start):.1f} max {(staged - start).max():.1f}")
print(f"row loop of a wave (us): min {(end - staged).min():.1f} median {np.median(end - staged):.1f} max {(end - staged).max():.1f}")
print(f"first-to-last wave end inside a workgroup (us): median {np.median(end.max(1) - end.min(1)):.1f} max {(end.max(1) - end.min(1)).max():.1f}")
print("mean wave end relative to the workgroup's first, by wave index:", np.round((end - end.min(1, keepdims=True)).mean(0), 1))
grid = np.linspace(0, end.max(), 33)
print("waves in their row loop / workgroups resident (any wave started, not all ended), over time:")
for a, b in zip(grid[:-1], grid[1:]):
    mid = (a + b) / 2
    nw = int(((staged <= mid) & (end > mid)).sum())
    ng = int(((wg_start <= mid) & (wg_end > mid)).sum())
    print(f"  t={mid:7.1f} us: waves {nw:5d} ({nw / 8192:.2f} of the chip's slots)  workgroups {ng:4d}")
cu, se, sh = (hw >> 8) & 0xF, (hw >> 13) & 0x7, (hw >> 12) & 1
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
ncu = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for k in np.unique(key)[:ncu]:
    m = np.nonzero(key == k)[0]
    for i in sorted(m, key=lambda i: wg_start[i]):
        print(f"  cu {k}: workgroup {i} (tile {i % ftiles}, chunk {i // ftiles}) waves start {start[i].min():.0f}..{start[i].max():.0f} end {end[i].min():.0f}..{end[i].max():.0f}")
cnt = np.bincount(key)
cnt = cnt[cnt > 0]
print(f"distinct CUs: {len(cnt)}; workgroups per CU: min {cnt.min()} max {cnt.max()}")
import os
if os.path.exists(sys.argv[1] + ".bt"):
    bt = np.fromfile(sys.argv[1] + ".bt", np.uint64).reshape(-1, 8).astype(np.int64)
    b0 = bt[:, 0].min()
    t = (bt - b0) / 100.0
    med = lambda x: float(np.median(x))
    print(f"bound_tables, {len(bt)} workgroups: span {t[:, 3].max():.1f} us; start median {med(t[:, 0]):.1f} max {t[:, 0].max():.1f}; "
          f"tables {med(t[:, 1] - t[:, 0]):.1f} (max {(t[:, 1] - t[:, 0]).max():.1f}); "
          f"scan {med(t[:, 2] - t[:, 1]):.1f} (max {(t[:, 2] - t[:, 1]).max():.1f}); "
          f"sort+merge {med(t[:, 3] - t[:, 2]):.1f} (max {(t[:, 3] - t[:, 2]).max():.1f})")
    print(f"  inside the table build (wave 0, after entry): sub-vector bounds {med(t[:, 4] - t[:, 0]):.1f}, first centroid group "
          f"{med(t[:, 5] - t[:, 0]):.1f}, last centroid group {med(t[:, 6] - t[:, 0]):.1f}, wave done {med(t[:, 7] - t[:, 0]):.1f}, "
          f"after the barrier {med(t[:, 1] - t[:, 0]):.1f}")
if os.path.exists(sys.argv[1] + ".qt"):
    qt = np.fromfile(sys.argv[1] + ".qt", np.uint64).reshape(-1, 4).astype(np.int64)
    t = (qt[:, :3] - qt[:, 0].min()) / 100.0
    print(f"qt_quantize, {len(qt)} workgroups: last end {t[:, 2].max():.1f} us; start median {np.median(t[:, 0]):.1f} max {t[:, 0].max():.1f}; "
          f"bounds ready after {np.median(t[:, 1] - t[:, 0]):.1f} (max {(t[:, 1] - t[:, 0]).max():.1f}); "
          f"loop {np.median(t[:, 2] - t[:, 1]):.1f} (max {(t[:, 2] - t[:, 1]).max():.1f})")
