#!/usr/bin/env python3
"""Dev probe: one synchronous az_propose per image (the harness's pattern) vs queue-ahead, ms per image."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "az-net_amd", "lib")); sys.path.insert(0, os.path.join(HERE, "..", ".."))
import torch
from aznet_hip import ffi, synth
from aznet_hip.net import HipAZNet
head = synth.make_head(seed=1234, **synth.FULL_DIMS)
net = HipAZNet(head, name="sync", max_regions=4096)
m = torch.from_numpy(synth.make_feature_map(5, 512, 38, 63)).cuda()
net.set_conv(m)
p = ffi.AzContext.make_params(600, 1000, 1.0, 0.0, static_tree=False)
for _ in range(20): net.propose(p)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): net.propose(p)
print("synchronous calls: %.4f ms per image" % ((time.perf_counter() - t0) / 200 * 1e3))
