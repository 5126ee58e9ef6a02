#!/bin/bash
# same-box A/B of the k-means assign stage at config 3: compact operand words against three pieces
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
timeout -k 10 900 python3 -m pytest tests/test_gpu_kmeans.py -x -q 2>&1 | tail -3 || exit 1
for i in 1 2; do
  for c in 1 0; do
    echo "compact=$c"; GULON_KMEANS_COMPACT=$c GULON_TRACE=1 python3 scripts/bench_kmeans.py 10000000 300 32 2 2>&1 | grep -E "assign stage1|re-checked|train " | tail -3
  done
done
