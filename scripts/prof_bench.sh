#!/bin/bash
# rocprofv3 kernel statistics of bench.py with extra arguments:  scripts/prof_bench.sh <name> [bench args...]
set -e
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$root/bench.py" --no-cpu-baseline --no-extras --no-recall "$@" > "$out.log" 2>&1
cd "$root"
python3 scripts/kstats.py "$out" > "$out.stats.txt"
head -30 "$out.stats.txt"
