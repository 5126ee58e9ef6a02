#!/bin/bash
# multi-word filter forms with compile-time word numbers: parity tests, then C5 / m = 25 / m = 32 benches
mkdir -p gpurun_out/r3_c5
timeout -k 10 600 python -m pytest tests/test_gpu_filter.py tests/test_gpu_baseline_configs.py tests/test_gpu_query.py tests/test_gpu_wide.py -x -q -m gpu > gpurun_out/r3_c5/tests.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r3_c5/tests.log | cut -c1-200
grep -n "Memory access fault\|Aborted\|Fatal" gpurun_out/r3_c5/tests.log | head -3
python bench.py --rows 10000000 --dim 1024 --quantizers 64 --steps 8 --warmup 2 --cpu-seconds 4 --no-extras 2>gpurun_out/r3_c5/c5.err | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C5  ms/step', round(r['ms_per_step'],3), r.get('parity_vs_oracle',{}).get('ids_equal'))"
python bench.py --rows 1000000 --dim 300 --quantizers 25 --steps 30 --warmup 5 --cpu-seconds 3 --no-extras 2>gpurun_out/r3_c5/m25.err | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('m25 ms/step', round(r['ms_per_step'],4), r.get('parity_vs_oracle',{}).get('ids_equal'))"
python bench.py --rows 4000000 --dim 96 --quantizers 32 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>gpurun_out/r3_c5/m32.err | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('m32 ms/step', round(r['ms_per_step'],4))"
tail -3 gpurun_out/r3_c5/m32.err | cut -c1-300
