#!/bin/bash
# same-box check of the k-means assign stage at config 3 (tests first, then the stage twice)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
timeout -k 10 900 python3 -m pytest tests/test_gpu_kmeans.py tests/test_gpu_baseline_configs.py -x -q 2>&1 | tail -3 || exit 1
for i in 1 2; do
  for lib in "" $1; do
    echo "lib=$lib"; GULON_HIP_LIB=$lib GULON_TRACE=1 python3 scripts/bench_kmeans.py 10000000 300 32 2 2>&1 | grep -E "assign stage1" | tail -2
  done
done
