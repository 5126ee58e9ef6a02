#!/bin/bash
# builds build/expt/libgulon_<tag>.so with extra compiler flags for some source files (timing experiments):
#   scripts/variant.sh <tag> <file-stem>[,<file-stem>...] <flags...>
#   e.g.  scripts/variant.sh glb4 filter,scan,conflict_order -DGULON_FILTER_GLB=4
set -e
tag=$1; stems=$2; shift 2
mkdir -p build/expt
for stem in ${stems//,/ }; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wall \
    -Wno-unused-function -Iinclude "$@" -c gulon_amd/csrc/$stem.hip -o build/expt/${stem}_$tag.o &
done
wait
objs=""
for f in api_core scan knn kmeans kmeans_stream kmeans_mfma replay filter grouped grouped_filter wide wide_filter conflict_order sharded literal; do
  if [[ ",$stems," == *",$f,"* ]]; then objs="$objs build/expt/${f}_$tag.o"; else objs="$objs build/obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/expt/libgulon_$tag.so $objs -ldl
echo build/expt/libgulon_$tag.so
