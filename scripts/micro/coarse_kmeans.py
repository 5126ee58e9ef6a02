import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gulon_amd as g
n, d, k = 10_000_000, 128, 10000
dm = g.DeviceMatrix.synthetic(n, d, 3, 1234, 1000)
t = time.perf_counter()
km = g.KMeans.compute_clusters(g.Vectors(dm), g.KMeansConfig(k, 1))
print("compute_clusters", time.perf_counter() - t, flush=True)
