#!/bin/bash
# stage times of the C3 k-means iteration (GULON_TRACE) for the in-tree library and build/expt variants:
#   scripts/c3_stage_times.sh [<tag> ...]
run() {
  GULON_TRACE=1 python3 scripts/bench_kmeans.py 10000000 300 32 3 2>&1 | grep "assign stage1\|update batch\|stages 2-3" | awk -v t="$1" '{a[$4" "$5]=a[$4" "$5]" "$(NF-1)} END{for(k in a) print t, k, a[k]}'
}
run base
for t in "$@"; do GULON_HIP_LIB=$PWD/build/expt/libgulon_$t.so run $t; done
