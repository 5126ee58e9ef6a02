#!/bin/bash
# Round 3 baseline of the 8-GPU strong-scaling point on one GPU: a 1.25 M-row shard through the multi-rank
# pipeline with a one-rank RCCL group (GULON_BENCH_REHEARSE=1): bench lines for 4 and 1 batches in flight and the
# kernel trace / statistics of both.   scripts/r3_shard_timeline.sh <tag> [rows]
set -e
tag=${1:-base}; rows=${2:-1250000}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3_$tag
mkdir -p "$out"
cd "$root"
export GULON_BENCH_REHEARSE=1
common="--rows $rows --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-recall"
for nfl in 4 1; do
  python3 bench.py $common --inflight $nfl > "$out/reh_nfl$nfl.json" 2> "$out/reh_nfl$nfl.err"
  echo "rehearsal rows=$rows inflight=$nfl: $(python3 -c "import json,sys; r=json.load(open('$out/reh_nfl$nfl.json')); print(r['ms_per_step'], r['roofline']['kernel_ms'])")"
done
cd /tmp && export TMPDIR=/tmp
for nfl in 4 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_nfl$nfl" -- python3 "$root/bench.py" --rows $rows --steps 40 --warmup 10 --no-cpu-baseline --no-extras --no-recall --inflight $nfl > "$out/prof_nfl$nfl.log" 2>&1
  python3 "$root/scripts/kstats.py" "$out/prof_nfl$nfl" > "$out/prof_nfl$nfl.stats.txt"
  python3 "$root/scripts/timeline.py" "$out/prof_nfl$nfl" > "$out/prof_nfl$nfl.timeline.txt" || true
  # keep the merge-back small: the raw trace is large
  find "$out/prof_nfl$nfl" -name '*kernel_trace.csv' -size +20M -delete || true
done
head -40 "$out/prof_nfl4.stats.txt"
