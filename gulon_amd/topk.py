"""TopKHeap.merge (TopKHeap.scala:44-53) on the device: merges per-shard partial
top-(K+1) lists under the deterministic (distance, row id) order."""
import numpy as np

from . import native as N


def merge_partials(part_dist, part_idx, k):
    """part_*: [lists][B][K+1] -> (idx [B][K], dist [B][K], count [B], flags [B])."""
    pd, pi = N.f32(part_dist), N.i32(part_idx)
    lists, b, keff = pd.shape
    if keff != k + 1:
        raise ValueError("partial lists must hold K+1 entries")
    oi = np.zeros((b, k), np.int32)
    od = np.zeros((b, k), np.float32)
    oc = np.zeros(max(b, 1), np.int32)
    of = np.zeros(max(b, 1), np.int32)
    N.check(N.lib().gulon_topk_merge(pd.reshape(-1), pi.reshape(-1), lists, b, k, oi.reshape(-1), od.reshape(-1),
                                     oc, of))
    return oi, od, oc[:b], of[:b]
