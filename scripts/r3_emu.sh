#!/bin/bash
# rank 0 of N on one GPU (GULON_BENCH_REHEARSE=N): bench lines for a few settings
#   scripts/r3_emu.sh <tag> [N] ; extra env through EMU_ENV="A=1 B=2"
set -e
tag=${1:-emu}; nranks=${2:-8}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3_$tag
mkdir -p "$out"
cd "$root"
run() {  # name, inflight, env...
  local name=$1 nfl=$2; shift 2
  env GULON_BENCH_REHEARSE=$nranks "$@" python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --inflight $nfl > "$out/$name.json" 2> "$out/$name.err" || { tail -5 "$out/$name.err"; return 1; }
  python3 -c "import json; r=json.load(open('$out/$name.json')); print('$name', 'ms/step', round(r['ms_per_step'],4), 'kernel', round(r['roofline']['kernel_ms'],4), 'recall', r.get('recall_at_10'), 'flagged', r.get('tie_flagged_queries'))"
}
run nfl4 4
run nfl1 1
run nfl6 6
run sample6700_nfl4 4 GULON_FILTER_SAMPLE=6700
run sample10000_nfl4 4 GULON_FILTER_SAMPLE=10000
run sample4096_nfl4 4 GULON_FILTER_SAMPLE=4096
