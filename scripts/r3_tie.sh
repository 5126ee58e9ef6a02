#!/bin/bash
# tie-heavy batches (data kind 1): replay level geometry variants
mkdir -p gpurun_out/r3_tie
for lib in ""; do
  for nfl in 0 1; do
    GULON_HIP_LIB=$lib python bench.py --steps 10 --warmup 3 --data-kind 1 --cpu-seconds 3 --no-extras --inflight $nfl 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('lib=$lib inflight=$nfl ms/step', round(r['ms_per_step'],4), r.get('parity_vs_oracle',{}).get('ids_equal'), r.get('parity_vs_oracle',{}).get('flagged_not_replayed'))"
  done
done
