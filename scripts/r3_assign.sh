#!/bin/bash
# k-means assign: parity tests, then the C3 stage times
set -e
mkdir -p gpurun_out/r3_assign
python -m pytest tests/test_gpu_kmeans.py tests/test_gpu_baseline_configs.py -x -q -m gpu > gpurun_out/r3_assign/tests.log 2>&1 || { tail -30 gpurun_out/r3_assign/tests.log; exit 1; }
tail -2 gpurun_out/r3_assign/tests.log
GULON_TRACE=1 python scripts/bench_kmeans.py 10000000 300 32 2 > gpurun_out/r3_assign/trace.log 2>&1
grep -E "assign stage1|stages 2-3|re-checked|update batch|train" gpurun_out/r3_assign/trace.log | tail -12
