#!/bin/bash
# HBM-side counters of the k-means placement kernel (one update at BASELINE config 3):
#   scripts/place_pmc.sh <tag>     (environment selects the variant, e.g. GULON_PLACE_STREAM=0)
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
export GULON_KMEANS_SERIAL=1
cd /tmp && export TMPDIR=/tmp
for grp in "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d "$out/$n" -- python3 $root/scripts/bench_kmeans.py 10000000 300 32 1 > "$out/$n.log" 2>&1 || echo "pass $n failed"
done
cd "$root"
python3 scripts/pmc_summary.py "sort_place" $out/* | sed "s/^/$tag /"
