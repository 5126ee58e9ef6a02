#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
mkdir -p $root/gpurun_out/r03; cd /tmp && export TMPDIR=/tmp
GULON_KMEANS_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r03/kmeans_pack -- python3 $root/scripts/bench_kmeans.py 10000000 300 32 1 > $root/gpurun_out/r03/kmeans_pack.log 2>&1
cd $root && python3 scripts/kstats.py gpurun_out/r03/kmeans_pack | head -6
find gpurun_out/r03/kmeans_pack -name '*kernel_trace.csv' -size +24M -delete
