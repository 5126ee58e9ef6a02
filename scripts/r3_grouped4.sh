#!/bin/bash
set -e
mkdir -p gpurun_out/r3_grouped
python -m pytest tests/test_gpu_grouped.py -x -q -m gpu > gpurun_out/r3_grouped/tests.log 2>&1 || { tail -40 gpurun_out/r3_grouped/tests.log; exit 1; }
tail -1 gpurun_out/r3_grouped/tests.log
for lib in ""; do
  GULON_HIP_LIB=$lib GULON_GROUPED_STATS=1 python tests/perf/bench_grouped.py 10000000 2>&1 >/dev/null | grep "by-group" | tail -1
  GULON_HIP_LIB=$lib python tests/perf/bench_grouped.py 10000000 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('lib=$lib 10M ms/step', r['ms_per_step'], r['parity_vs_oracle'])"
  GULON_HIP_LIB=$lib python tests/perf/bench_grouped.py 1000000 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('lib=$lib 1M ms/step', r['ms_per_step'], r['parity_vs_oracle'])"
done
