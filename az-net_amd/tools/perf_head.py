#!/usr/bin/env python3
"""Per-stage timing of one head forward (RoIPool, fc6/fc7 GEMM + reduce, epilogue) for a
range of roi counts, from HIP events on the ctx stream.  Development tool."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "lib"))
from aznet_hip import ffi, synth            # noqa: E402
from aznet_hip.net import HipAZNet          # noqa: E402


def main():
    Rs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 8, 32, 64, 130, 517, 1024, 2048]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    head = synth.make_head(seed=1234, **synth.FULL_DIMS)
    net = HipAZNet(head, max_regions=4096)          # AZ_GEMM_MODE=0|2|3 selects the int6 kernel
    fmap = synth.make_feature_map(4, 512, 38, 63)
    net.set_conv(fmap)
    rng = np.random.RandomState(0)
    flop6 = 2 * 25088 * 4096
    for R in Rs:
        x1 = rng.uniform(0, 900, R); y1 = rng.uniform(0, 500, R)
        rois = np.stack([np.zeros(R), x1, y1, x1 + rng.uniform(16, 99, R), y1 + rng.uniform(16, 99, R)], 1).astype(np.float32)
        net.ctx.head_forward(rois)
        net.ctx.set_profiling(2 | 4)
        for _ in range(reps):
            net.ctx.head_forward(rois)
        t = net.ctx.last_kernel_times()
        net.ctx.set_profiling(0)
        agg = {}
        for n, l, ms in t:
            agg.setdefault(n, []).append(ms)
        med = {k: float(np.median(v)) * 1e3 for k, v in agg.items()}
        tf = R * flop6 / (med["fc6_gemm"] * 1e-6) / 1e12
        bw = 411041792 / (med["fc6_gemm"] * 1e-6) / 1e12
        print("R=%5d  " % R + "  ".join("%s %7.1f" % (k, med[k]) for k in
              ("roi_pool", "fc6_gemm", "fc6_reduce", "fc7_gemm", "tail")) +
              "  | fc6 %.1f TF/s  W-stream %.2f TB/s" % (tf, bw))


if __name__ == "__main__":
    main()
