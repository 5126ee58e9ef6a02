#!/bin/bash
# usage: scripts/kres.sh <file.hip> [name filter] -- per-kernel VGPR/SGPR/scratch/LDS/occupancy summary
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Iinclude ${KRES_FLAGS} -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | python3 -c "
import re, subprocess, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ''
cur, rows = None, []
for line in sys.stdin:
    m = re.search(r'remark: +(.*?) +\[-Rpass', line)
    if not m: continue
    t = m.group(1)
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1); cur[k.strip()] = v.strip()
names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r'\(anonymous namespace\)::|gulon::|void ', '', n)
    n = re.sub(r'\(.*', '', n)
    if flt and flt not in n: continue
    print(f\"{n[:70]:70s} SGPR {r.get('TotalSGPRs','?'):>4s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s} LDS {r.get('LDS Size [bytes/block]','?')}\")
" "$2"
