#!/bin/bash
# HBM / L2 counters of the grouped-index query kernels (one pass per counter group)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r03
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  n=pmc_gf_$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d "$out/$n" -- python3 $root/tests/perf/bench_grouped.py 10000000 > "$out/$n.log" 2>&1 || { echo "pass failed: $grp"; tail -5 "$out/$n.log"; }
done
cd "$root"
for k in "gf_filter<16>" gq_cdist gq_select_groups "gf_quant<16>" "gf_survivors<16>"; do
  python3 scripts/pmc_summary.py "$k" $out/pmc_gf_* > "$out/grouped_${k%%<*}_pmc.csv"; echo "== $k"; cat "$out/grouped_${k%%<*}_pmc.csv"
done
find "$out" -path "*pmc_gf_*" -name '*counter_collection.csv' -size +8M -delete
