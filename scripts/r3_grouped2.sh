#!/bin/bash
# grouped index at 10 M rows: step time, fallback counts, kernel statistics
set -e
mkdir -p gpurun_out/r3_grouped
GULON_GROUPED_STATS=1 python tests/perf/bench_grouped.py 10000000 > gpurun_out/r3_grouped/new_10M.json 2> gpurun_out/r3_grouped/new_10M.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/r3_grouped/new_10M.json").read().strip().split("\n")[-1])
print("10M ms/step (with stats syncs)", r["ms_per_step"], "recall", r["recall_at_10"], r["parity_vs_oracle"])
PY
grep "approximate pre-selection\|by-group" gpurun_out/r3_grouped/new_10M.err | sort | uniq -c | tail -3
python tests/perf/bench_grouped.py 10000000 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('10M ms/step', r['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3_grouped/prof -o g10 -- python3 $GRAFT_REPO_ROOT/tests/perf/bench_grouped.py 10000000 > $GRAFT_REPO_ROOT/gpurun_out/r3_grouped/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/kstats.py gpurun_out/r3_grouped/prof 2>/dev/null | grep -i "gq_\|gf_\|merge" | head -20
