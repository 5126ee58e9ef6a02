set -e
mkdir -p gpurun_out/dyn
timeout -k 10 1100 python -m pytest tests/test_gpu_filter.py tests/test_gpu_shared_bounds.py tests/test_gpu_query.py tests/test_gpu_sharded_build.py -x -q > gpurun_out/dyn/test.log 2>&1 || { tail -40 gpurun_out/dyn/test.log; exit 1; }
tail -2 gpurun_out/dyn/test.log
for r in 1250000 10000000; do for f in 1 2; do
 GULON_BENCH_INFLIGHT=$f python bench.py --rows $r --no-cpu-baseline --steps 60 2>/dev/null > gpurun_out/dyn/${r}_$f.json
done; done
python bench.py --rows 1000000 --no-cpu-baseline --steps 100 2>/dev/null > gpurun_out/dyn/c2.json
timeout -k 10 300 python tests/perf/bench_shared_bounds.py 8 > gpurun_out/dyn/emul_8.json 2> gpurun_out/dyn/emul_8.err; cat gpurun_out/dyn/emul_8.json
