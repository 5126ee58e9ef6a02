#!/bin/bash
# the headline bench (no CPU leg, no extras) for the in-tree library and for each build/expt/libgulon_<tag>.so given:
#   scripts/bench_variants.sh [--inflight N] <tag> [<tag> ...]
extra=""
if [ "$1" = "--inflight" ]; then extra="--inflight $2"; shift 2; fi
run() {
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-recall $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), 'q/s', round(d['ms_per_step'],3), 'ms/step  kernel', round(d['roofline']['kernel_ms'],3))"
}
run base
for t in "$@"; do GULON_HIP_LIB=$PWD/build/expt/libgulon_$t.so run $t; done
