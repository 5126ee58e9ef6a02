#!/bin/bash
# Round-3 profiles, collected on the GPU box into gpurun_out/r03/ (summaries are copied to profiles/r03/):
#   scripts/collect_profiles_r03.sh [part ...]     parts: bench pmc exact shard forms kmeans kind1 other grouped
# Every rocprofv3 invocation has the python program directly after `--`; counters in passes of their own
# (FETCH_SIZE and WRITE_SIZE never share a pass), never together with tracing other than --kernel-trace.
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r03
mkdir -p "$out" "$out/sharded"
parts=${@:-bench pmc exact shard forms kmeans kind1 other}
cd /tmp && export TMPDIR=/tmp
stats() {  # name, program args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$name" -- "$@" > "$out/$name.log" 2>&1
  (cd "$root" && python3 scripts/kstats.py "$out/$name" > "$out/$name.kernel_stats.txt")
  find "$out/$name" -name '*kernel_trace.csv' -size +24M -delete
  echo "== $name"; head -6 "$out/$name.kernel_stats.txt"
}
pmc() {    # name, counters (space separated), program args...
  local name=$1 ctrs=$2; shift 2
  rocprofv3 --pmc $ctrs --output-format csv -d "$out/$name" -- "$@" > "$out/$name.log" 2>&1
}
filter_pmc() {  # tag, kernel name, bench args...
  local tag=$1 kern=$2; shift 2
  for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
    n=pmc_${tag}_$(echo $grp | tr ' ' '_' | cut -c1-40)
    pmc $n "$grp" python3 $root/bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-recall --no-extras "$@"
  done
  (cd "$root" && python3 scripts/pmc_summary.py "$kern" $out/pmc_${tag}_* > "$out/${tag}_kernel_pmc.csv")
  echo "== $tag counters"; cat "$out/${tag}_kernel_pmc.csv"
  find "$out" -path "*pmc_${tag}_*" -name '*counter_collection.csv' -size +8M -delete
}
for part in $parts; do case $part in
bench)
  (cd "$root" && python3 bench.py --steps 20 --warmup 3 > "$out/bench_n1_with_extras.json" 2> "$out/bench_n1_with_extras.err")
  stats bench python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-extras
  ;;
pmc)
  filter_pmc filter "filter_kernel<16, 1, 16, 4, 1, 2>"
  ;;
exact)
  GULON_SCAN_FILTER=0 stats bench_exact_scan python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-recall --no-extras
  ;;
shard)
  (cd "$root"
   for nfl in 4 1; do GULON_BENCH_REHEARSE=8 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --inflight $nfl > "$out/sharded/rank0_of_8_inflight$nfl.json" 2>/dev/null; done
   for nfl in 4 1; do GULON_BENCH_REHEARSE=1 python3 bench.py --rows 1250000 --steps 200 --warmup 20 --no-cpu-baseline --no-extras --inflight $nfl > "$out/sharded/rehearsal_1rank_1250000_inflight$nfl.json" 2>/dev/null; done
   python3 bench.py --rows 1250000 --steps 200 --warmup 20 --no-cpu-baseline --no-extras > "$out/sharded/plain_1250000.json" 2>/dev/null
   GULON_HOST_PROFILE=1 GULON_BENCH_REHEARSE=8 python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-recall --inflight 4 2>&1 >/dev/null | grep "\[host\]" > "$out/sharded/rank0_of_8_host_profile.txt")
  GULON_BENCH_REHEARSE=8 stats sharded/rank0_of_8_nfl1 python3 $root/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras --no-recall --inflight 1
  (cd "$root" && python3 scripts/timeline.py "$out/sharded/rank0_of_8_nfl1" > "$out/sharded/rank0_of_8_timeline.txt")
  GULON_BENCH_REHEARSE=8 stats sharded/rank0_of_8_nfl4 python3 $root/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras --no-recall --inflight 4
  ;;
forms)
  (cd "$root"
   python3 bench.py --rows 10000000 --dim 1024 --quantizers 64 --steps 10 --warmup 2 --cpu-seconds 10 --no-extras > "$out/bench_c5_10Mx1024_m64.json" 2>/dev/null
   python3 bench.py --rows 1000000 --dim 300 --quantizers 25 --steps 30 --warmup 5 --cpu-seconds 5 --no-extras > "$out/bench_cli_default_1Mx300_m25.json" 2>/dev/null
   python3 bench.py --rows 1000000 --steps 50 --warmup 5 --cpu-seconds 5 --no-extras > "$out/bench_c2_1M.json" 2>/dev/null
   python3 bench.py --rows 4000000 --dim 96 --quantizers 32 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$out/bench_4Mx96_m32.json" 2>/dev/null)
  filter_pmc c5 "filter_kernel<8, 1, 16, 2, 1, 0>" --rows 10000000 --dim 1024 --quantizers 64
  filter_pmc m25 "filter_kernel<16, 1, 4, 2, 1, 0>" --rows 1000000 --dim 300 --quantizers 25
  ;;
kmeans)
  GULON_KMEANS_SERIAL=1 stats kmeans_c3 python3 $root/scripts/bench_kmeans.py 10000000 300 32 2
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
    n=pmc_kmeans_$(echo $grp | tr ' ' '_' | cut -c1-30)
    GULON_KMEANS_SERIAL=1 pmc $n "$grp" python3 $root/scripts/bench_kmeans.py 10000000 300 32 1
  done
  (cd "$root" && for k in assign_bf16 stream_chains stream_order; do python3 scripts/pmc_summary.py "$k" $out/pmc_kmeans_* > "$out/kmeans_c3_${k}_pmc.csv"; done)
  cat "$out/kmeans_c3_assign_bf16_pmc.csv"
  (cd "$root" && GULON_TRACE=1 python3 scripts/bench_kmeans.py 10000000 300 32 2 > "$out/kmeans_c3_trace.txt" 2>&1); tail -16 "$out/kmeans_c3_trace.txt"
  find "$out" -path "*pmc_kmeans_*" -name '*counter_collection.csv' -size +8M -delete
  ;;
kind1)
  stats bench_kind1 python3 $root/bench.py --steps 5 --warmup 2 --inflight 1 --data-kind 1 --no-cpu-baseline --no-recall --no-extras
  ;;
other)
  (cd "$root" && python3 tests/perf/bench_grouped.py 10000000 2>/dev/null | tail -1 > "$out/bench_grouped_10M.json"; python3 tests/perf/bench_grouped.py 1000000 2>/dev/null | tail -1 > "$out/bench_grouped_1M.json"
   python3 tests/perf/bench_wide.py 2>/dev/null | tail -1 > "$out/bench_wide_1M_k1024.json"; python3 tests/perf/bench_wide.py 1000000 4096 2>/dev/null | tail -1 > "$out/bench_wide_1M_k4096.json")
  ;;
grouped)
  stats grouped_10M python3 $root/tests/perf/bench_grouped.py 10000000
  ;;
esac; done
echo done
