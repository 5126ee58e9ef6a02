set -e
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/glb3
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/glb3/test.log 2>&1 || { tail -40 gpurun_out/glb3/test.log; exit 1; }
tail -2 gpurun_out/glb3/test.log
for rep in 1 2; do for v in prev new; do
 if [ $v = prev ]; then export GULON_HIP_LIB=$R/build/prev/libgulon_prev.so; else unset GULON_HIP_LIB; fi
 python bench.py --no-cpu-baseline --steps 40 2>/dev/null > gpurun_out/glb3/${v}_full_$rep.json
 python bench.py --rows 1250000 --no-cpu-baseline --no-recall --steps 200 2>/dev/null > gpurun_out/glb3/${v}_s_$rep.json
 python tests/perf/bench_shared_bounds.py 8 > gpurun_out/glb3/emul_${v}_$rep.json 2>/dev/null
 python bench.py --rows 1000000 --no-cpu-baseline --no-recall --steps 200 2>/dev/null > gpurun_out/glb3/${v}_c2_$rep.json
done; done
