"""BASELINE config 3: KMeans codebook training, 10M x 300, m = 32 (sub-dims 12x10 + 20x9), k = 256.
Reports build time and per-iteration time; run under rocprofv3 for per-kernel numbers.
    python scripts/bench_kmeans.py [n] [d] [m] [iters]"""
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gulon_amd as g

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 300
m = int(sys.argv[3]) if len(sys.argv) > 3 else 32
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
k = 256
t0 = time.perf_counter()
dm = g.DeviceMatrix.synthetic(n, d, 2, 1234, 1)
t1 = time.perf_counter()
reps = []
pq = g.ProductQuantizer.apply(dm, g.ProductQuantizerConfig(k, m, iters, reps.append))
t2 = time.perf_counter()
nrep = max(len(r) for r in reps[0])
passes = nrep                      # assign passes: init + one per report after the first
flops = 2.0 * n * k * d * passes
print(f"n={n} d={d} m={m} k={k} maxIterations={iters}: synth {t1-t0:.2f}s train {t2-t1:.3f}s "
      f"({passes} assign passes, {(t2-t1)/passes*1e3:.1f} ms per full-PQ iteration incl. update, "
      f"{flops/(t2-t1)/1e12:.1f} TFLOP/s end-to-end)", flush=True)
t3 = time.perf_counter()
enc = pq.encode(dm)
print(f"encode {time.perf_counter()-t3:.3f}s", flush=True)
