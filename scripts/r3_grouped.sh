#!/bin/bash
# grouped index: parity tests, then the 10 M / 1 M benches with and without the by-group filter
set -e
mkdir -p gpurun_out/r3_grouped
python -m pytest tests/test_gpu_grouped.py -x -q -m gpu > gpurun_out/r3_grouped/tests.log 2>&1 || { tail -40 gpurun_out/r3_grouped/tests.log; exit 1; }
tail -2 gpurun_out/r3_grouped/tests.log
for n in 1000000 10000000; do
  GULON_GROUPED_STATS=1 python tests/perf/bench_grouped.py $n > gpurun_out/r3_grouped/new_$n.json 2> gpurun_out/r3_grouped/new_$n.err
  tail -1 gpurun_out/r3_grouped/new_$n.json; grep "approximate pre-selection" gpurun_out/r3_grouped/new_$n.err | sort | uniq -c | tail -3
  GULON_GROUPED_FILTER=0 python tests/perf/bench_grouped.py $n > gpurun_out/r3_grouped/old_$n.json 2> gpurun_out/r3_grouped/old_$n.err
  tail -1 gpurun_out/r3_grouped/old_$n.json
done
