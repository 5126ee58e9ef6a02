#!/bin/bash
# A/B of the k-means update kernels at BASELINE config 3 under rocprofv3 (one kernel at a time):
#   scripts/upd_ab.sh <tag> [iters]      -> gpurun_out/<tag>.stats.txt
set -e
tag=$1; iters=${2:-3}
export GULON_KMEANS_SERIAL=1
scripts/prof.sh "$tag" scripts/bench_kmeans.py 10000000 300 32 "$iters" > /dev/null
grep -E "update_chains|sort_place|sort_hist|sort_scan" gpurun_out/$tag.stats.txt
python3 - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/{sys.argv[1]}/*/*kernel_trace.csv")[0]
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "update_chains" in n or "sort_place" in n:
        print(n.split("(")[0][-40:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
