#!/bin/bash
# counters of the k-means assign kernel at BASELINE config 3 (one sub-quantizer launch at a time)
#   scripts/r3_assign_pmc.sh <tag> [kernel-name-substring]
tag=${1:-assign}; kern=${2:-assign_bf16}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r3_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  GULON_KMEANS_SERIAL=1 rocprofv3 --pmc $grp --output-format csv -d "$out/pmc$i" -- python3 "$root/scripts/bench_kmeans.py" 10000000 300 32 1 > "$out/pmc$i.log" 2>&1 || tail -3 "$out/pmc$i.log"
done
cd "$root"
python3 scripts/pmc_summary.py "$kern" $out/pmc* | tee "$out/${kern}_pmc.csv"
find "$out" -name '*counter_collection.csv' -size +8M -delete
