#!/bin/bash
# Stage timing of the single-workgroup geometry kernels (k_spec_levels, k_level_geom): builds a copy of the library with
# AZ_FUSED_TIMING / AZ_LEVEL_TIMING (device printf of wall_clock64 stamps, x10 ns) and runs a few searches.
# usage (GPU box, repo root): bash az-net_amd/tools/stage_timing.sh [Tz]
set -e
repo=$(pwd)
b=/tmp/az_timing_build
rm -rf $b && mkdir -p $b/csrc $b/include $b/lib/aznet_hip
cp az-net_amd/csrc/*.hip az-net_amd/csrc/*.h az-net_amd/csrc/Makefile $b/csrc/
mkdir -p $b/../include_dummy
# the sources include ../../include/aznet_hip.h relative to csrc
mkdir -p /tmp/include && cp include/aznet_hip.h /tmp/include/
sed -i 's#\.\./\.\./include/aznet_hip.h#/tmp/include/aznet_hip.h#' $b/csrc/az_dev.h $b/csrc/Makefile
make -s -C $b/csrc OUT=$b/lib/aznet_hip/libaznet_hip.so CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DAZ_FUSED_TIMING -DAZ_LEVEL_TIMING -DAZ_SPEC_TIMING" 2>&1 | grep -E "error" || true
python3 - "$b/lib/aznet_hip/libaznet_hip.so" "${1:-0.0}" <<'PY'
import sys, os
sys.path.insert(0, "az-net_amd/lib"); sys.path.insert(0, ".")
from aznet_hip import ffi
ffi.load_library(sys.argv[1])
ffi._lib = ffi.load_library(sys.argv[1])
from aznet_hip import synth
from aznet_hip.net import HipAZNet
head = synth.make_head(seed=1234, **synth.FULL_DIMS)
net = HipAZNet(head, max_regions=4096)
net.set_conv(synth.make_feature_map(4, 512, 38, 63))
Tz = float(sys.argv[2])
p = ffi.AzContext.make_params(600, 1000, 1.0, Tz, static_tree=False)
for i in range(4):
    print("---- search", i, flush=True)
    Y, st = net.propose(p, want_stats=True)
    print("passes", list(st.pass_rows[:st.n_passes]), flush=True)
PY
