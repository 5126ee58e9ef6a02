#!/bin/bash
# bench lines of the other filter forms and of the headline (after the committed counters were re-stamped)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r03
mkdir -p "$out"; cd "$root"
python3 bench.py --rows 10000000 --dim 1024 --quantizers 64 --steps 10 --warmup 2 --cpu-seconds 10 --no-extras > "$out/bench_c5_10Mx1024_m64.json" 2>/dev/null
python3 bench.py --rows 1000000 --dim 300 --quantizers 25 --steps 30 --warmup 5 --cpu-seconds 5 --no-extras > "$out/bench_cli_default_1Mx300_m25.json" 2>/dev/null
python3 bench.py --steps 20 --warmup 3 > "$out/bench_n1_with_extras.json" 2> "$out/bench_n1_with_extras.err"
for f in bench_c5_10Mx1024_m64 bench_cli_default_1Mx300_m25 bench_n1_with_extras; do python3 -c "
import json,sys; d=json.load(open('$out/$f.json')); r=d['roofline']; print('$f', d['ms_per_step'], r['frac'], r.get('physical_note'), (r.get('physical') or {}).keys())"; done
