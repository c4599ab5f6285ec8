"""The kernel table of the headline search (HIP events per launch group, 8 untimed steps) -- a quick A/B probe:
AZNET_HIP_LIB=<variant> python tests/dev/kt_quick.py"""
import sys
sys.path.insert(0, "az-net_amd/lib"); sys.path.insert(0, ".")
import numpy as np, torch
from aznet_hip import ffi, synth
from aznet_hip.net import HipAZNet
head = synth.make_head(seed=1234, **synth.FULL_DIMS)
net = HipAZNet(head, name="kt", max_regions=4096)
maps = [torch.from_numpy(synth.make_feature_map(s, 512, 38, 63)).cuda().contiguous(memory_format=torch.channels_last) for s in range(3)]
prm = ffi.AzContext.make_params(600, 1000, 1.0, 0.0, static_tree=False)
for rows, p in (("688-row whole-tree pass", prm),):
    for i in range(30):
        net.ctx.propose_launch(p, fmap=maps[i % 3], producer_done=True); net.ctx.propose_fetch()
    net.ctx.set_profiling(2 | 4)
    for i in range(8):
        net.ctx.propose_launch(p, fmap=maps[i % 3], producer_done=True); net.ctx.propose_fetch()
    kt = net.ctx.last_kernel_times(); net.ctx.set_profiling(0)
    by = {}
    for nm, l, ms in kt: by.setdefault(nm, []).append(ms * 1e3)
    print(rows, {k: round(float(np.mean(v)), 1) for k, v in by.items()})
# deep tree (config 4)
fm = torch.from_numpy(synth.make_feature_map(4, 512, 38, 57)).cuda().contiguous(memory_format=torch.channels_last)
p4 = ffi.AzContext.make_params(800, 1200, 0.75, 0.0, static_tree=False)
for i in range(6):
    net.ctx.propose_launch(p4, fmap=fm, producer_done=True); net.ctx.propose_fetch()
net.ctx.set_profiling(2 | 4)
for i in range(4):
    net.ctx.propose_launch(p4, fmap=fm, producer_done=True); net.ctx.propose_fetch()
kt = net.ctx.last_kernel_times(); net.ctx.set_profiling(0)
by = {}
for nm, l, ms in kt: by.setdefault(nm, []).append(ms * 1e3)
print("config 4", {k: round(float(np.sum(v)) / 4, 1) for k, v in by.items()})
