#!/bin/bash
# tie-heavy batches (data kind 1): replay level geometry variants
mkdir -p gpurun_out/r3_tie
for lib in "" build/expt/libgulon_l1_1024.so build/expt/libgulon_l1_512.so build/expt/libgulon_l1_1024s32.so; do
  for nfl in 0 1; do
    GULON_HIP_LIB=$lib python bench.py --steps 10 --warmup 3 --data-kind 1 --no-cpu-baseline --no-extras --inflight $nfl 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('lib=$lib inflight=$nfl ms/step', round(r['ms_per_step'],4), 'recall', r.get('recall_at_10'), r.get('parity_vs_oracle',{}).get('ids_equal'))"
  done
done
