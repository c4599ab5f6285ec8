"""Stage stamps of the geometry kernels on a pruned tree (objects map at the tuned threshold) and on the full tree.
AZNET_HIP_LIB=<a build with -DAZ_FUSED_TIMING -DAZ_LEVEL_TIMING> python tests/dev/geom_timing.py"""
import sys
sys.path.insert(0, "az-net_amd/lib"); sys.path.insert(0, ".")
import torch
from aznet_hip import ffi, synth
from aznet_hip.net import HipAZNet
ohead = synth.make_object_head(seed=1234, **synth.FULL_DIMS)
net = HipAZNet(ohead, name="gt", max_regions=4096)
m = torch.from_numpy(synth.make_object_map(3, 512, 38, 63)).cuda().contiguous(memory_format=torch.channels_last)
for tz in (0.51, 0.0):
    p = ffi.AzContext.make_params(600, 1000, 1.0, tz, static_tree=False)
    for i in range(4):
        print("--- Tz", tz, "search", i, flush=True)
        net.ctx.propose_launch(p, fmap=m, producer_done=True)
        Y, st = net.ctx.propose_fetch(want_stats=True)
        print("regions", [int(st.level_regions[l]) for l in range(st.n_levels)], "passes", list(st.pass_rows[:st.n_passes]), flush=True)
