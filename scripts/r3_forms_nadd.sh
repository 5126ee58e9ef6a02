#!/bin/bash
# the multi-word filter forms with 7-bit (NADD 2, their default) and 6-bit levels (NADD 4)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3_forms_nadd; mkdir -p "$out"; cd "$root"
line() { python3 -c "import json; r=json.load(open('$1')); print('$2', 'ms/step', round(r['ms_per_step'],4), 'kernel', round(r['roofline']['kernel_ms'],4), 'recall', r.get('recall_at_10'))"; }
for nadd in 2 4; do
  GULON_FILTER_NADD=$nadd python bench.py --rows 10000000 --dim 1024 --quantizers 64 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$out/c5_nadd$nadd.json" 2>/dev/null; line "$out/c5_nadd$nadd.json" "c5_m64 nadd=$nadd"
  GULON_FILTER_NADD=$nadd python bench.py --rows 1000000 --dim 300 --quantizers 25 --steps 30 --warmup 5 --no-cpu-baseline --no-extras > "$out/m25_nadd$nadd.json" 2>/dev/null; line "$out/m25_nadd$nadd.json" "cli_m25 nadd=$nadd"
  GULON_FILTER_NADD=$nadd python bench.py --rows 4000000 --dim 96 --quantizers 32 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$out/m32_nadd$nadd.json" 2>/dev/null; line "$out/m32_nadd$nadd.json" "m32 nadd=$nadd"
done
