#!/bin/bash
# plain single index of <rows> rows with environment knobs:  scripts/r3_plain_knobs.sh <tag> <rows> "<name> <nfl> ENV=.." ...
tag=$1; rows=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3_$tag
mkdir -p "$out"
cd "$root"
for spec in "$@"; do
  set -- $spec
  name=$1; nfl=$2; shift 2
  env "$@" python3 bench.py --rows $rows --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-recall --inflight $nfl > "$out/$name.json" 2> "$out/$name.err" || { tail -3 "$out/$name.err"; continue; }
  python3 -c "import json; r=json.load(open('$out/$name.json')); print('$name', 'ms/step', round(r['ms_per_step'],4), 'kernel', round(r['roofline']['kernel_ms'],4))"
done
