#!/bin/bash
# rank 0 of 8 with environment knobs:  scripts/r3_emu_knobs.sh <tag> "<name> <nfl> ENV=.. ENV=.." ...
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3_$tag
mkdir -p "$out"
cd "$root"
for spec in "$@"; do
  set -- $spec
  name=$1; nfl=$2; shift 2
  env GULON_BENCH_REHEARSE=8 "$@" python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-recall --inflight $nfl > "$out/$name.json" 2> "$out/$name.err" || { tail -3 "$out/$name.err"; continue; }
  python3 -c "import json; r=json.load(open('$out/$name.json')); print('$name', 'ms/step', round(r['ms_per_step'],4), 'kernel', round(r['roofline']['kernel_ms'],4))"
done
