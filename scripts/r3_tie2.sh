#!/bin/bash
# tie replay with level 1 through the filter: parity tests, tie-heavy and default benches with and without it
set -e
mkdir -p gpurun_out/r3_tie
python -m pytest tests -x -q -m gpu -k "tie or replay or query or filter or shard or golden" > gpurun_out/r3_tie/tests.log 2>&1 || { tail -30 gpurun_out/r3_tie/tests.log; exit 1; }
tail -1 gpurun_out/r3_tie/tests.log
for l1 in 1 0; do
  for nfl in 0 1; do
    GULON_REPLAY_L1_FILTER=$l1 python bench.py --steps 10 --warmup 3 --data-kind 1 --cpu-seconds 4 --no-extras --inflight $nfl 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('tie-heavy l1_filter=$l1 inflight=$nfl ms/step', round(r['ms_per_step'],4), r.get('parity_vs_oracle'))"
  done
  GULON_REPLAY_L1_FILTER=$l1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('default   l1_filter=$l1 ms/step', round(r['ms_per_step'],4))"
done
