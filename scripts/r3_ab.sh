#!/bin/bash
# same-box A/B of the headline bench: this tree's library against build/expt/libgulon_old8ca.so (commit 8ca616e)
for i in 1 2 3; do
  for lib in "" build/expt/libgulon_old8ca.so; do
    GULON_HIP_LIB=$lib python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-recall 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('lib=$lib', round(r['ms_per_step'],4), round(r['roofline']['kernel_ms'],4))"
  done
done
