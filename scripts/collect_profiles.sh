#!/bin/bash
# Round profiles, collected on the GPU box into gpurun_out/<tag>/ (copy the summaries to profiles/<tag>/):
#   scripts/collect_profiles.sh <tag>
# Every rocprofv3 invocation has the python program directly after `--`; counters in passes of their own
# (FETCH_SIZE and WRITE_SIZE never share a pass: TCC slots), never together with tracing other than --kernel-trace.
set -e
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py"
stats() {  # name, program args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$name" -- "$@" > "$out/$name.log" 2>&1
  (cd "$root" && python3 scripts/kstats.py "$out/$name" > "$out/$name.kernel_stats.txt")
  echo "== $name"; head -8 "$out/$name.kernel_stats.txt"
}
pmc() {    # name, counters (space separated), program args...
  local name=$1 ctrs=$2; shift 2
  rocprofv3 --pmc $ctrs --output-format csv -d "$out/$name" -- "$@" > "$out/$name.log" 2>&1
}
# 1. the headline bench: kernel statistics, then the main-stage filter kernel's counters
stats bench python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-extras
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES"; do
  n=pmc_filter_$(echo $grp | tr ' ' '_' | cut -c1-40)
  pmc $n "$grp" python3 $root/bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-recall --no-extras
done
(cd "$root" && python3 scripts/pmc_summary.py "filter_kernel<16, 1, 16, 4, 1, 2>" $out/pmc_filter_* > "$out/filter_kernel_pmc.csv")
cat "$out/filter_kernel_pmc.csv"
# 2. the exact scan (filter off)
GULON_SCAN_FILTER=0 stats bench_exact_scan python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-recall --no-extras
# 3. BASELINE config 3: k-means, kernels one at a time (GULON_KMEANS_SERIAL=1), then the assign / update counters
GULON_KMEANS_SERIAL=1 stats kmeans_c3 python3 $root/scripts/bench_kmeans.py 10000000 300 32 2
GULON_KMEANS_SERIAL=1 pmc pmc_kmeans "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_MFMA" python3 $root/scripts/bench_kmeans.py 10000000 300 32 1
(cd "$root" && python3 scripts/pmc_summary.py "assign_bf16" $out/pmc_kmeans > "$out/kmeans_c3_assign_pmc.csv"; python3 scripts/pmc_summary.py "update_chains" $out/pmc_kmeans > "$out/kmeans_c3_chains_pmc.csv"; python3 scripts/pmc_summary.py "sort_place" $out/pmc_kmeans > "$out/kmeans_c3_place_pmc.csv")
cat "$out/kmeans_c3_assign_pmc.csv"
(cd "$root" && GULON_TRACE=1 python3 scripts/bench_kmeans.py 10000000 300 32 2 > "$out/kmeans_c3_trace.txt" 2>&1); tail -22 "$out/kmeans_c3_trace.txt"
# 4. GroupedIndex, 10 M rows
stats grouped_10M python3 $root/tests/perf/bench_grouped.py 10000000
(cd "$root" && python3 tests/perf/bench_grouped.py 10000000 2>/dev/null | tail -1 > "$out/bench_grouped_10M.json"; python3 tests/perf/bench_grouped.py 1000000 2>/dev/null | tail -1 > "$out/bench_grouped_1M.json")
# 5. wide codes (k = 1024) through the quantized filter, and the reference-shaped data (every query an exact tie)
stats wide_1M_k1024 python3 $root/tests/perf/bench_wide.py
(cd "$root" && python3 tests/perf/bench_wide.py 2>/dev/null | tail -1 > "$out/bench_wide_1M_k1024.json"; GULON_SCAN_FILTER=0 python3 tests/perf/bench_wide.py 2>/dev/null | tail -1 > "$out/bench_wide_1M_k1024_exact.json")
stats bench_kind1 python3 $root/bench.py --steps 5 --warmup 2 --inflight 1 --data-kind 1 --no-cpu-baseline --no-recall --no-extras
echo done
