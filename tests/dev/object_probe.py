#!/usr/bin/env python3
"""Builder's probe: zoom-score populations of the planted-object set at the FULL head (hot anchors vs the rest), trees at the tuned Tz."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "az-net_amd", "lib")); sys.path.insert(0, REPO)
import torch
from aznet_hip import ffi, synth
from aznet_hip.net import HipAZNet
H, W = 600, 1000
for kw in (dict(), dict(noise=0.0), dict(beta=0.2)):
    mo = kw.pop("max_objects", 4)
    head = synth.make_object_head(seed=1234, **dict(synth.FULL_DIMS, **kw))
    net = HipAZNet(head, name="probe", max_regions=4096)
    maps = [synth.make_object_map(j, 512, 38, 63, max_objects=mo) for j in range(16)]
    net.ctx.tune_begin(16 * 2 * 4096)
    allz = []
    for m in maps:
        net.set_conv(m)
        net.propose(ffi.AzContext.make_params(H, W, 1.0, 0.0, tune=True))
        B, z = net.ctx.last_anchors()
        allz.append(z)
    tz = net.ctx.tune_kth_largest(16 * 20)[0]
    net.ctx.tune_end()
    z = np.concatenate(allz)
    print(kw, "max_objects", mo, "Tz@20", tz, "quantiles 50/90/99/99.5/99.9:", np.round(np.quantile(z, [0.5, 0.9, 0.99, 0.995, 0.999]), 3),
          "per image > 0.5:", [int((a > 0.5).sum()) for a in allz[:8]])
    for i, m in enumerate(maps[:10]):
        net.set_conv(m)
        Y, st = net.propose(ffi.AzContext.make_params(H, W, 1.0, tz), want_stats=True)
        print("   ", i, [int(st.level_regions[l]) for l in range(st.n_levels)], int((m[0, 0] > 0).sum()))
    del net
