#!/bin/bash
# streamed k-means update: parity tests, then the C3 stage times and the kernel statistics
set -e
mkdir -p gpurun_out/r3_stream
python -m pytest tests/test_gpu_kmeans.py -x -q -m gpu > gpurun_out/r3_stream/tests.log 2>&1 || { tail -30 gpurun_out/r3_stream/tests.log; exit 1; }
tail -2 gpurun_out/r3_stream/tests.log
GULON_TRACE=1 python scripts/bench_kmeans.py 10000000 300 32 2 > gpurun_out/r3_stream/trace.log 2>&1
grep -E "update batch|assign|iteration|total" gpurun_out/r3_stream/trace.log | tail -12
cd /tmp && export TMPDIR=/tmp
GULON_KMEANS_SERIAL=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3_stream/prof -o km -- python $GRAFT_REPO_ROOT/scripts/bench_kmeans.py 10000000 300 32 2 > $GRAFT_REPO_ROOT/gpurun_out/r3_stream/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/kernel_stats.py gpurun_out/r3_stream/prof 2>/dev/null | head -12 || true
