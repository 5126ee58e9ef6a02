#!/bin/bash
# kernel timeline of one headline batch with a single batch in flight (what every kernel costs alone)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r03
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/bench_nfl1" -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-recall --inflight 1 > "$out/bench_nfl1.log" 2>&1
cd "$root" && python3 scripts/timeline.py "$out/bench_nfl1" > "$out/bench_nfl1_timeline.txt"
find "$out/bench_nfl1" -name '*kernel_trace.csv' -size +24M -delete
cat "$out/bench_nfl1_timeline.txt"
