#!/bin/bash
# gfx950 ISA of one kernel of an object in build/obj:  scripts/disasm.sh <stem> <mangled-name substring> > out.s
set -e
stem=$1; pat=$2
tmp=$(mktemp -d)
cp build/obj/$stem.o $tmp/
(cd $tmp && /opt/rocm/lib/llvm/bin/llvm-objdump -d --offloading $stem.o > /dev/null 2>&1 || true)
dev=$(ls $tmp | grep gfx950 | head -1)
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $tmp/$dev | awk -v pat="$pat" '
  /^[0-9a-f]+ <.*>:$/ { on = index($0, pat) > 0 }
  on { print }'
rm -rf $tmp
