#!/bin/bash
# grouped index at 10 M rows: sample size sweep (threshold quality against sample cost)
set -e
mkdir -p gpurun_out/r3_grouped
python -m pytest tests/test_gpu_grouped.py -x -q -m gpu > gpurun_out/r3_grouped/tests.log 2>&1 || { tail -40 gpurun_out/r3_grouped/tests.log; exit 1; }
tail -1 gpurun_out/r3_grouped/tests.log
for sg in 4; do
  GULON_GROUPED_SAMPLE=$sg GULON_GROUPED_STATS=1 python tests/perf/bench_grouped.py 10000000 2>&1 >/dev/null | grep "by-group" | tail -1
  GULON_GROUPED_SAMPLE=$sg python tests/perf/bench_grouped.py 10000000 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('sample groups $sg: 10M ms/step', r['ms_per_step'], r['parity_vs_oracle'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3_grouped/prof2 -o g10 -- python3 $GRAFT_REPO_ROOT/tests/perf/bench_grouped.py 10000000 > $GRAFT_REPO_ROOT/gpurun_out/r3_grouped/prof2.log 2>&1
cd $GRAFT_REPO_ROOT
if [ -f build/expt/libgulon_gfst.so ]; then
  GULON_HIP_LIB=build/expt/libgulon_gfst.so python tests/perf/bench_grouped.py 10000000 2>/dev/null | grep "gf_filter tile" | sort | uniq -c | sort -rn | head -40 > gpurun_out/gfst.log || true
fi
