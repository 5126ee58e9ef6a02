set -e
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/glb2
for v in libgulon_prev glb2 glb3; do
 GULON_HIP_LIB=$R/build/prev/$v.so python bench.py --rows 1250000 --no-cpu-baseline --no-recall --steps 200 2>/dev/null > gpurun_out/glb2/${v}_s.json
 GULON_HIP_LIB=$R/build/prev/$v.so python bench.py --rows 1000000 --no-cpu-baseline --no-recall --steps 200 2>/dev/null > gpurun_out/glb2/${v}_c2.json
 GULON_HIP_LIB=$R/build/prev/$v.so python bench.py --rows 4000000 --dim 96 --quantizers 32 --no-cpu-baseline --no-recall --steps 30 2>/dev/null > gpurun_out/glb2/${v}_m32.json
 GULON_HIP_LIB=$R/build/prev/$v.so python bench.py --rows 2000000 --dim 256 --quantizers 64 --no-cpu-baseline --no-recall --steps 30 2>/dev/null > gpurun_out/glb2/${v}_m64.json
 GULON_HIP_LIB=$R/build/prev/$v.so python bench.py --rows 1000000 --dim 300 --quantizers 25 --no-cpu-baseline --no-recall --steps 60 2>/dev/null > gpurun_out/glb2/${v}_m25.json
 GULON_HIP_LIB=$R/build/prev/$v.so python bench.py --rows 1000000 --dim 200 --quantizers 100 --no-cpu-baseline --no-recall --steps 30 2>/dev/null > gpurun_out/glb2/${v}_m100.json
done
