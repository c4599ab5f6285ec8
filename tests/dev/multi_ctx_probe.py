#!/usr/bin/env python3
"""Builder's probe: the object stream on N independent contexts (one lane each, round-robin) -- how much do more searches in
flight buy in the sparse, latency-bound regime?"""
import os, sys, time, gc
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "az-net_amd", "lib")); sys.path.insert(0, REPO)
import torch
from aznet_hip import ffi, synth
from aznet_hip.net import HipAZNet
H, W = 600, 1000
dense = "--dense" in sys.argv
head = synth.make_object_head(seed=1234, **synth.FULL_DIMS) if not dense else synth.make_head(seed=1234, **synth.FULL_DIMS)
n_img = 32
maps = [torch.from_numpy(synth.make_object_map(j, 512, 38, 63)).cuda().contiguous(memory_format=torch.channels_last) for j in range(n_img)]
nets = [HipAZNet(head, name="p%d" % i, max_regions=4096) for i in range(8)]
net = nets[0]
net.ctx.tune_begin(n_img * 2 * 4096)
for m in maps:
    net.set_conv(m); net.propose(ffi.AzContext.make_params(H, W, 1.0, 0.0, tune=True))
tz = net.ctx.tune_kth_largest(n_img * 20)[0]
net.ctx.tune_end()
if dense: tz = 0.0
prm = ffi.AzContext.make_params(H, W, 1.0, tz)
gc.collect(); gc.disable()
for ncx, lanes in ((1, 2), (2, 1), (3, 1), (4, 1), (6, 1), (8, 1), (2, 2), (3, 2), (4, 2)):
    for n in nets[:ncx]: n.ctx.set_lanes(lanes)
    def run(k):
        q = []
        cap = ncx * (lanes + 1 if lanes == 2 else 1)
        for i in range(k):
            n = nets[(i // (1 if lanes == 1 else 1)) % ncx]
            if len(q) == cap:
                q.pop(0).ctx.propose_fetch()
            n.ctx.propose_launch(prm, fmap=maps[i % n_img], producer_done=True)
            q.append(n)
        for m in q: m.ctx.propose_fetch()
    run(2 * n_img)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(6 * n_img)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("contexts %d x lanes %d: %.4f ms/image" % (ncx, lanes, dt / (6 * n_img) * 1e3))
