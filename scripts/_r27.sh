set -e
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/final3
python bench.py > gpurun_out/final3/bench_n1.json 2> gpurun_out/final3/bench_n1.err
cat gpurun_out/final3/bench_n1.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final3/stats -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/final3/bench_n1_under_rocprof.json 2> $R/gpurun_out/final3/rocprof.err
GULON_BENCH_INFLIGHT=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/final3/trace1 -- python3 $R/bench.py --no-cpu-baseline --no-recall --steps 5 > $R/gpurun_out/final3/bench_inflight1.json 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final3/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-recall --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final3/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-recall --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/final3/pmc_sq -- python3 $R/bench.py --no-cpu-baseline --no-recall --steps 3 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $R/gpurun_out/final3/pmc_ta -- python3 $R/bench.py --no-cpu-baseline --no-recall --steps 3 --warmup 1 > /dev/null 2>&1 || echo "ta counters unavailable"
