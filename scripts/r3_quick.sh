#!/bin/bash
# quick shard check: bench lines (rehearsal, 4 and 1 in flight) + the kernel timeline of one batch
#   scripts/r3_quick.sh <tag> [rows] [extra bench args...]
set -e
tag=${1:-q}; rows=${2:-1250000}; shift; shift || true
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3_$tag
mkdir -p "$out"
cd "$root"
export GULON_BENCH_REHEARSE=${GULON_BENCH_REHEARSE:-1}
common="--rows $rows --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-recall"
for nfl in 4 1; do
  python3 bench.py $common --inflight $nfl "$@" > "$out/reh_nfl$nfl.json" 2> "$out/reh_nfl$nfl.err"
  echo "$tag rows=$rows inflight=$nfl: $(python3 -c "import json,sys; r=json.load(open('$out/reh_nfl$nfl.json')); print(round(r['ms_per_step'],4), round(r['roofline']['kernel_ms'],4))")"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_nfl1" -- python3 "$root/bench.py" --rows $rows --steps 40 --warmup 10 --no-cpu-baseline --no-extras --no-recall --inflight 1 "$@" > "$out/prof_nfl1.log" 2>&1
python3 "$root/scripts/kstats.py" "$out/prof_nfl1" > "$out/prof_nfl1.stats.txt"
python3 "$root/scripts/timeline.py" "$out/prof_nfl1" > "$out/prof_nfl1.timeline.txt" || true
find "$out/prof_nfl1" -name '*kernel_trace.csv' -size +20M -delete || true
head -12 "$out/prof_nfl1.timeline.txt"
