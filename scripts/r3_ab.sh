#!/bin/bash
# same-box A/B of the headline bench: this tree's library against the libraries named
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root"
for i in 1 2 3; do
  for lib in "" "$@"; do
    GULON_HIP_LIB=$lib python3 bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-recall 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('lib=$lib', round(r['ms_per_step'],4), round(r['roofline']['kernel_ms'],4))"
  done
done
