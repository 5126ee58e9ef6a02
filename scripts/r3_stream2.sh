#!/bin/bash
# streamed k-means update: parity tests, stage times, section stamps of the chain kernel
set -e
mkdir -p gpurun_out/r3_stream
python -m pytest tests/test_gpu_kmeans.py -x -q -m gpu > gpurun_out/r3_stream/tests.log 2>&1 || { tail -30 gpurun_out/r3_stream/tests.log; exit 1; }
tail -2 gpurun_out/r3_stream/tests.log
GULON_TRACE=1 python scripts/bench_kmeans.py 10000000 300 32 2 > gpurun_out/r3_stream/trace.log 2>&1
grep -E "update batch|total|train" gpurun_out/r3_stream/trace.log | tail -5
if [ -f build/expt/libgulon_sstamps.so ]; then
  GULON_HIP_LIB=build/expt/libgulon_sstamps.so python scripts/bench_kmeans.py 10000000 300 32 1 > gpurun_out/sstamps.log 2>&1
  grep "wg (0,0)" gpurun_out/sstamps.log | tail -4
fi
