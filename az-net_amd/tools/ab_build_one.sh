#!/bin/bash
# A/B build of libaznet_hip.so: recompile ONE source with extra -D flags, link with the tree's other objects.
#   az-net_amd/tools/ab_build_one.sh <tag> <source.hip> [-DNAME=VALUE ...]   ->  az-net_amd/csrc/dev/ab/libaznet_hip_<tag>.so
# (run with AZNET_HIP_LIB=<that file>; built files travel with gpurun, are not tracked)
set -eu
here=$(cd "$(dirname "$0")/../csrc" && pwd)
tag=$1; src=$2; shift 2
mkdir -p "$here/dev/ab"
obj="$here/dev/ab/${src%.hip}_$tag.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -I"$here" "$@" -c "$here/$src" -o "$obj"
others=$(ls "$here"/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$here/dev/ab/libaznet_hip_$tag.so" $others "$obj" -ldl
echo "$here/dev/ab/libaznet_hip_$tag.so"
