#!/usr/bin/env python3
"""Accuracy of the split-bf16 int6 path (gemm_mode 2) vs the fp32-MFMA path and the BLAS oracle,
plus row-independence across batch sizes.  Checker script (imports the oracle, so it lives under tests/; not collected by pytest)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "az-net_amd", "lib"))
sys.path.insert(0, os.path.join(HERE, ".."))
from aznet_hip import synth              # noqa: E402
from aznet_hip.net import HipAZNet       # noqa: E402
from oracle import az_oracle as orc      # noqa: E402

small = len(sys.argv) > 1 and sys.argv[1] == "small"
dims = synth.SMALL_DIMS if small else synth.FULL_DIMS
head = synth.make_head(seed=1234, **dims)
fmap = synth.make_feature_map(4, dims["C"], 38, 63)
rng = np.random.RandomState(3)
R = 300
x1 = rng.uniform(0, 900, R); y1 = rng.uniform(0, 500, R)
rois = np.stack([np.zeros(R), x1, y1, x1 + rng.uniform(16, 300, R), y1 + rng.uniform(16, 300, R)], 1).astype(np.float32)
zr, pr, dr = orc.head_forward(head, fmap[0], rois)
outs = {}
for mode in (0, 2):
    net = HipAZNet(head, max_regions=1024, gemm_mode=mode)
    net.set_conv(fmap)
    for n in (1, 40, 64, 65, 130, 300):
        z, p, d = net.ctx.head_forward(rois[:n])
        outs[(mode, n)] = (z, p, d)
        print("mode %d R=%3d vs BLAS: zoom %.2e prob %.2e delta %.2e | rows equal to R=300 run: %s" % (
            mode, n, np.abs(z - zr[:n]).max(), np.abs(p - pr[:n]).max(), np.abs(d - dr[:n]).max(),
            "-" if (mode, 300) not in outs else all(np.array_equal(a, b[:n]) for a, b in zip((z, p, d), outs[(mode, 300)]))))
    z3 = outs[(mode, 300)]
    for n in (1, 40, 64, 65, 130):
        print("   mode %d R=%3d rows == R=300 rows: %s" % (mode, n, all(np.array_equal(a, b[:n]) for a, b in zip(outs[(mode, n)], z3))))
    del net
