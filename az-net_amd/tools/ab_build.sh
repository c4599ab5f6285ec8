#!/bin/bash
# Build a variant of the library with extra compiler flags into /tmp/az_ab_<name>/libaznet_hip.so (A/B measurements on
# the GPU box: AZNET_HIP_LIB=/tmp/az_ab_<name>/libaznet_hip.so python bench.py ...).
# usage: bash az-net_amd/tools/ab_build.sh <name> "<extra flags>"
set -e
name=$1; flags=$2
b=/tmp/az_ab_$name
rm -rf $b && mkdir -p $b/csrc
cp az-net_amd/csrc/*.hip az-net_amd/csrc/*.h az-net_amd/csrc/Makefile $b/csrc/
mkdir -p /tmp/include && cp include/aznet_hip.h /tmp/include/
sed -i 's#\.\./\.\./include/aznet_hip.h#/tmp/include/aznet_hip.h#' $b/csrc/az_dev.h $b/csrc/Makefile
make -s -j4 -C $b/csrc OUT=$b/libaznet_hip.so CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $flags" 2>&1 | grep -E " error" || true
ls -la $b/libaznet_hip.so
