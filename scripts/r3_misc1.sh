#!/bin/bash
# full GPU suite, the other filter forms, lane on/off at three sizes
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3_misc1
mkdir -p "$out"; cd "$root"
python -m pytest tests -m gpu -x -q > "$out/tests.log" 2>&1; tail -2 "$out/tests.log"
line() { python3 -c "import json; r=json.load(open('$1')); print('$2', 'ms/step', round(r['ms_per_step'],4), 'kernel', round(r['roofline']['kernel_ms'],4), 'recall', r.get('recall_at_10'))"; }
python bench.py --rows 10000000 --dim 1024 --quantizers 64 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$out/c5.json" 2> "$out/c5.err"; line "$out/c5.json" c5_m64
python bench.py --rows 1000000 --dim 300 --quantizers 25 --steps 30 --warmup 5 --no-cpu-baseline --no-extras > "$out/m25.json" 2> "$out/m25.err"; line "$out/m25.json" cli_m25
python bench.py --rows 4000000 --dim 96 --quantizers 32 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$out/m32.json" 2> "$out/m32.err"; line "$out/m32.json" m32
python bench.py --rows 1000000 --steps 50 --warmup 5 --no-cpu-baseline --no-extras > "$out/c2.json" 2> "$out/c2.err"; line "$out/c2.json" c2_1M
for lane in 1 0; do
  GULON_FILTER_LANE=$lane python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-recall > "$out/n10m_lane$lane.json" 2>/dev/null; line "$out/n10m_lane$lane.json" "10M lane=$lane"
  GULON_FILTER_LANE=$lane python bench.py --rows 1250000 --steps 200 --warmup 10 --no-cpu-baseline --no-extras --no-recall > "$out/n125_lane$lane.json" 2>/dev/null; line "$out/n125_lane$lane.json" "1.25M lane=$lane"
  GULON_FILTER_LANE=$lane GULON_BENCH_REHEARSE=8 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extras --no-recall --inflight 4 > "$out/emu8_lane$lane.json" 2>/dev/null; line "$out/emu8_lane$lane.json" "emu8 lane=$lane"
done
