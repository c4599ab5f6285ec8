#!/usr/bin/env python3
"""Dev probe: a 'dataset' that mixes image shapes (full head, Tz = 0, queue-ahead): ms per image when the shapes alternate
against the same shapes one after the other."""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "az-net_amd", "lib")); sys.path.insert(0, os.path.join(HERE, "..", ".."))
import torch
from aznet_hip import ffi, synth
from aznet_hip.net import HipAZNet
head = synth.make_head(seed=1234, **synth.FULL_DIMS)
net = HipAZNet(head, name="mixed", max_regions=4096)
shapes = [(600, 1000, 1.0), (375, 500, 1.6), (480, 640, 1.25), (500, 375, 1.6)]
maps = [torch.from_numpy(synth.make_feature_map(50 + i, 512, synth.conv_out_size(int(round(H * sc))),
                                                synth.conv_out_size(int(round(W * sc))))).cuda() for i, (H, W, sc) in enumerate(shapes)]
prm = [ffi.AzContext.make_params(H, W, sc, 0.0, static_tree=False) for (H, W, sc) in shapes]


def run(order, n):
    net.ctx.propose_launch(prm[order[0]], fmap=maps[order[0]], producer_done=True)
    for i in range(n):
        if i + 1 < n:
            j = order[(i + 1) % len(order)]
            net.ctx.propose_launch(prm[j], fmap=maps[j], producer_done=True)
        net.ctx.propose_fetch()


for name, order in (("one shape at a time", None), ("alternating", [0, 1, 2, 3])):
    if order is None:
        tot = 0.0
        for i in range(len(shapes)):
            run([i], 20)
            torch.cuda.synchronize(); t0 = time.perf_counter(); run([i], 100); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 100 * 1e3
            print("   shape %s: %.3f ms" % (shapes[i], dt)); tot += dt
        print("%s: mean %.3f ms per image" % (name, tot / len(shapes)))
    else:
        run(order, 40)
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(order, 400); torch.cuda.synchronize()
        print("%s: %.3f ms per image" % (name, (time.perf_counter() - t0) / 400 * 1e3))
