#!/bin/bash
# rocprofv3 kernel statistics of one python command on the GPU box:
#   scripts/prof.sh <name> <script.py> [args...]      -> gpurun_out/<name>/ + gpurun_out/<name>.stats.txt
# (the program goes directly after `--`: no env/bash hops under the profiler)
set -e
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$name
mkdir -p "$out"
script=$root/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$script" "$@" > "$out.log" 2>&1
cd "$root"
python3 scripts/kstats.py "$out" > "$out.stats.txt"
head -40 "$out.stats.txt"
