#!/usr/bin/env python3
"""Dev probe: int6 arithmetic modes (az_set_gemm_mode 0 / 2 / 3) -- error of the head outputs against an f64
evaluation of the same head, row independence, and time per head pass / per search.  Checker script (imports the
oracle for RoIPool), not collected by pytest."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "az-net_amd", "lib"))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from aznet_hip import ffi, synth         # noqa: E402
from aznet_hip.net import HipAZNet       # noqa: E402
from oracle import az_oracle as orc      # noqa: E402

modes = [int(m) for m in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,2,3".split(","))]
dims = synth.FULL_DIMS
head = synth.make_head(seed=1234, **dims)
fmap = synth.make_feature_map(4, dims["C"], 38, 63)
rng = np.random.RandomState(3)
R = 700
x1 = rng.uniform(0, 900, R); y1 = rng.uniform(0, 500, R)
rois = np.stack([np.zeros(R), x1, y1, x1 + rng.uniform(16, 300, R), y1 + rng.uniform(16, 300, R)], 1).astype(np.float32)


def f64_head(n):
    p5 = orc.roi_pool(fmap[0], rois[:n]).reshape(n, -1).astype(np.float64)
    f = lambda x, W, b, relu: (np.maximum(x @ W.astype(np.float64).T + b, 0) if relu else x @ W.astype(np.float64).T + b)
    h6 = f(p5, head["W6"], head["b6"], True)
    h71 = f(h6, head["W71"], head["b71"], True)
    h72 = f(h6, head["W72"], head["b72"], True)
    sg = lambda x: 1.0 / (1.0 + np.exp(-x))
    return sg(f(h72, head["Wz"], head["bz"], False)), sg(f(h71, head["Was"], head["bas"], False)), f(h71, head["Wab"], head["bab"], False)


fast = len(sys.argv) > 2 and sys.argv[2] == "fast"
NT = 8 if fast else 200
truth = f64_head(NT)
zr, pr, dr = orc.head_forward(head, fmap[0], rois[:NT])
print("fp32 BLAS oracle vs f64: zoom %.2e prob %.2e delta %.2e" % tuple(np.abs(a - b).max() for a, b in zip((zr, pr, dr), truth)))
for mode in modes:
    net = HipAZNet(head, max_regions=4096, gemm_mode=mode)
    net.set_conv(fmap)
    out = net.ctx.head_forward(rois)
    print("mode %d vs f64 (first %d rows): zoom %.2e prob %.2e delta %.2e   (mean abs delta err %.2e)" % (
        (mode, NT) + tuple(np.abs(a[:NT] - b).max() for a, b in zip(out, truth)) + (np.abs(out[2][:NT] - truth[2]).mean(),)))
    for n in ((65,) if fast else (1, 40, 64, 65, 130, 300)):
        sub = net.ctx.head_forward(rois[:n])
        ok = all(np.array_equal(a, b[:n]) for a, b in zip(sub, out))
        if not ok:
            print("   mode %d: rows of a %d-row launch differ from the %d-row launch" % (mode, n, R))
    # time per head pass at a few row counts (HIP events around each launch group)
    net.ctx.set_profiling(2 | 4)
    for n in (48, 160, 670):
        for _ in range(3):
            net.ctx.head_forward(rois[:n])
        net.ctx.last_kernel_times()
        for _ in range(10):
            net.ctx.head_forward(rois[:n])
        t = {}
        for name, l, ms in net.ctx.last_kernel_times():
            t.setdefault(name, []).append(ms)
        print("   mode %d rows %3d: " % (mode, n) + "  ".join("%s %.1f us" % (k, 1e3 * np.mean(v)) for k, v in t.items()))
    net.ctx.set_profiling(0)
    for Tz, static in ((0.0, False), (0.0, True)):
        p = ffi.AzContext.make_params(600, 1000, 1.0, Tz, static_tree=static)
        for _ in range(10):
            net.propose(p)
        t0 = time.perf_counter()
        for _ in range(100):
            net.propose(p)
        print("   mode %d search (%s): %.3f ms" % (mode, "one pass" if static else "level loop", (time.perf_counter() - t0) * 10))
    del net
