#!/bin/bash
# same-box A/B of the k-means assign stage at config 3: this tree's library against the libraries named
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
for i in 1 2; do
  for lib in "" "$@"; do
    echo "lib=$lib $(GULON_HIP_LIB=$lib GULON_TRACE=1 python3 scripts/bench_kmeans.py 10000000 300 32 2 2>&1 | grep -E "assign stage1" | tail -2 | awk '{print $5}' | tr '\n' ' ')"
  done
done
