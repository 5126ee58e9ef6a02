#!/bin/bash
# usage: scripts/kres.sh <file.hip>  -- per-kernel VGPR/SGPR/scratch/LDS/occupancy summary
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Iinclude -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|ScratchSize|LDS Size|Occupancy|SGPRs:|AGPRs" \
 | sed -E 's/^.*remark: +//; s/ +\[-Rpass.*//; s/Function Name: /\n/' | tr '\n' '\t' | sed 's/\t\t/\n/g' | c++filt | sed -E 's/\(.*\)//' | cut -c1-300
echo
