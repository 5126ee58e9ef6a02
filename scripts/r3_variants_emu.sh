#!/bin/bash
# rank 0 of 8 (and the 10 M-row single GPU line) for the in-tree library and build/expt variants:
#   scripts/r3_variants_emu.sh <tag> <variant>...
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3_$tag
mkdir -p "$out"
cd "$root"
line() { python3 -c "import json; r=json.load(open('$1')); print('$2', 'ms/step', round(r['ms_per_step'],4), 'kernel', round(r['roofline']['kernel_ms'],4), 'recall', r.get('recall_at_10'))"; }
for v in base "$@"; do
  lib=""; [ "$v" != base ] && lib="GULON_HIP_LIB=$root/build/expt/libgulon_$v.so"
  for nfl in 4 1; do
    env $lib GULON_BENCH_REHEARSE=8 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --inflight $nfl > "$out/emu8_${v}_nfl$nfl.json" 2> "$out/emu8_${v}_nfl$nfl.err" && line "$out/emu8_${v}_nfl$nfl.json" "emu8 $v nfl$nfl"
  done
  for nfl in 3 1; do
    env $lib python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-recall --inflight $nfl > "$out/n1_${v}_nfl$nfl.json" 2> "$out/n1_${v}_nfl$nfl.err" && line "$out/n1_${v}_nfl$nfl.json" "10M $v nfl$nfl"
  done
done
