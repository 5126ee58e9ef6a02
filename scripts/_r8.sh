set -e
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/fb2
timeout -k 10 1100 python -m pytest tests/test_gpu_filter.py tests/test_gpu_shared_bounds.py tests/test_gpu_query.py -x -q > gpurun_out/fb2/test.log 2>&1 || { tail -40 gpurun_out/fb2/test.log; exit 1; }
tail -2 gpurun_out/fb2/test.log
for rep in 1 2; do for v in prev new; do
 if [ $v = prev ]; then export GULON_HIP_LIB=$R/build/prev/libgulon_prev.so; else unset GULON_HIP_LIB; fi
 python bench.py --no-cpu-baseline --no-recall --steps 60 2>/dev/null > gpurun_out/fb2/${v}_full_$rep.json
 python bench.py --rows 1250000 --no-cpu-baseline --no-recall --steps 200 2>/dev/null > gpurun_out/fb2/${v}_1250000_$rep.json
done; done
unset GULON_HIP_LIB
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fb2/stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/fb2/bench_n1_under_rocprof.json 2> $R/gpurun_out/fb2/rocprof.err
