#!/bin/bash
# per-workgroup stamps of the main-stage filter kernel on a shard:  scripts/r3_stamps.sh <tag> [rows] [extra env...]
set -e
tag=${1:-stamps}; rows=${2:-1250000}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3_$tag
mkdir -p "$out"
cd "$root"
GULON_HIP_LIB=$root/build/expt/libgulon_stamps.so GULON_FILTER_STAMPS=$out/stamps.bin python3 bench.py --rows $rows --steps 3 --warmup 1 \
  --inflight 1 --no-cpu-baseline --no-extras --no-recall > "$out/bench.json" 2> "$out/bench.err"
python3 scripts/stamps_report.py "$out/stamps.bin" | tee "$out/stamps.txt"
