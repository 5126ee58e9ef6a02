#!/bin/bash
# headline step against the bound cascade's knobs: sample rows / extra first stage / second stage (blocks per 128)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"; mkdir -p gpurun_out
run() { echo "$* -> $(env "$@" python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-recall 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))")"; }
run GULON_FILTER_SAMPLE=65536
run GULON_FILTER_SAMPLE=32768
run GULON_FILTER_SAMPLE=16384
run GULON_FILTER_SAMPLE=16384 GULON_FILTER_STAGE0=1
run GULON_FILTER_SAMPLE=8192 GULON_FILTER_STAGE0=1
run GULON_FILTER_SAMPLE=16384 GULON_FILTER_STAGE0=1 GULON_FILTER_STAGE1=8
run GULON_FILTER_SAMPLE=32768 GULON_FILTER_STAGE1=8
run GULON_FILTER_SAMPLE=65536 GULON_FILTER_STAGE1=8
run GULON_FILTER_SAMPLE=65536 GULON_FILTER_STAGE1=12
run GULON_FILTER_SAMPLE=65536 GULON_FILTER_STAGE1=14
run GULON_FILTER_SAMPLE=16384 GULON_FILTER_STAGE0=2 GULON_FILTER_STAGE1=12
run GULON_FILTER_SAMPLE=65536
