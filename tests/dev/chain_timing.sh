#!/bin/bash
# Stage stamps of the geometry kernels (printf from the kernels; builds with -DAZ_FUSED_TIMING -DAZ_LEVEL_TIMING)
bash az-net_amd/tools/ab_build.sh tim "-DAZ_FUSED_TIMING -DAZ_LEVEL_TIMING" > /dev/null 2>&1
AZNET_HIP_LIB=/tmp/az_ab_tim/libaznet_hip.so python - <<'PY'
import os, sys
sys.path.insert(0, "az-net_amd/lib"); sys.path.insert(0, ".")
from aznet_hip import ffi, synth
from aznet_hip.net import HipAZNet
head = synth.make_head(seed=77, **synth.SMALL_DIMS)
net = HipAZNet(head, name="tim")
net.set_conv(synth.make_feature_map(5, synth.SMALL_DIMS["C"], 38, 63))
p = ffi.AzContext.make_params(600, 1000, 1.0, 0.0, static_tree=False)
for i in range(4):
    print("--- search", i, flush=True)
    Y, st = net.propose(p, want_stats=True)
    print("passes", list(st.pass_rows[:st.n_passes]), flush=True)
PY
