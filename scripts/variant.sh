#!/bin/bash
# builds build/expt/libgulon_<tag>.so with extra compiler flags for ONE source file (timing experiments):
#   scripts/variant.sh <tag> <file-stem> <flags...>      e.g.  scripts/variant.sh glb4 filter -DGULON_FILTER_GLB=4
set -e
tag=$1; stem=$2; shift 2
mkdir -p build/expt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wall \
  -Wno-unused-function -Iinclude "$@" -c gulon_amd/csrc/$stem.hip -o build/expt/${stem}_$tag.o
objs=""
for f in api_core scan knn kmeans kmeans_fused kmeans_mfma replay filter grouped wide wide_filter sharded literal; do
  if [ $f = $stem ]; then objs="$objs build/expt/${stem}_$tag.o"; else objs="$objs build/obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/expt/libgulon_$tag.so $objs -ldl
echo build/expt/libgulon_$tag.so
