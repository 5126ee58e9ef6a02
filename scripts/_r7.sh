set -e
mkdir -p gpurun_out/smp
for smp in 65536 12288 8192 4096; do
 GULON_FILTER_SAMPLE=$smp python tests/perf/bench_shared_bounds.py 8 > gpurun_out/smp/emul8_$smp.json 2>/dev/null
done
python tests/perf/bench_shared_bounds.py 4 > gpurun_out/smp/emul4.json 2>/dev/null
