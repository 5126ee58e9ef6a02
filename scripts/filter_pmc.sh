#!/bin/bash
# Counters of the main-stage filter kernel of the headline bench (one batch in flight), one pass per group:
#   scripts/filter_pmc.sh <tag> [kernel name]   -> gpurun_out/<tag>/filter_kernel_pmc.csv
set -e
tag=${1:-fpmc}
kern=${2:-"filter_kernel<16, 1, 16, 4, 1, 2>"}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES"; do
  n=pmc_filter_$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d "$out/$n" -- python3 $root/bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-recall --no-extras > "$out/$n.log" 2>&1
done
cd "$root"
python3 scripts/pmc_summary.py "$kern" $out/pmc_filter_* > "$out/filter_kernel_pmc.csv"
cat "$out/filter_kernel_pmc.csv"
