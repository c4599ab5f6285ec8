#!/usr/bin/env python3
"""Dev probe: launch groups of the calibrated-Tz search (config A, Tz = median zoom score of levels 2-3), HIP events."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "az-net_amd", "lib")); sys.path.insert(0, os.path.join(HERE, "..", ".."))
from aznet_hip import ffi, synth
from aznet_hip.net import HipAZNet
head = synth.make_head(seed=1234, **synth.FULL_DIMS)
net = HipAZNet(head, name="calib", max_regions=4096)
fmap = synth.make_feature_map(31, 512, 38, 63)
net.set_conv(fmap)
net.propose(ffi.AzContext.make_params(600, 1000, 1.0, 0.0, tune=True))
zz = net.ctx.last_anchors()[1].astype(np.float64)
for q in (0.5, 0.3):
    tz = float(np.quantile(zz[1:41], q))
    p = ffi.AzContext.make_params(600, 1000, 1.0, tz)
    for _ in range(5):
        Y, st = net.propose(p, want_stats=True)
    net.ctx.set_profiling(2 | 4)
    for _ in range(10):
        net.propose(p)
    t = {}
    for n, l, ms in net.ctx.last_kernel_times():
        t.setdefault((n, l), []).append(ms)
    net.ctx.set_profiling(0)
    print("Tz q%.1f: regions %s unique %s passes %s" % (q, list(st.level_regions[:5]), list(st.level_unique[:5]), list(st.pass_rows[:st.n_passes])))
    tot = 0
    for (n, l), v in sorted(t.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        print("   %-14s L%-2d %7.1f us" % (n, l + 1, 1e3 * np.mean(v))); tot += 1e3 * np.mean(v)
    print("   sum %.1f us" % tot)
