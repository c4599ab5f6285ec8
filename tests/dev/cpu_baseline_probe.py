#!/usr/bin/env python3
"""Builder's probe: the CPU baseline's configuration on the GPU box's host -- NumPy BLAS vs torch addmm at several thread counts."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "az-net_amd", "lib")); sys.path.insert(0, REPO)
import torch
from threadpoolctl import threadpool_limits, threadpool_info
from aznet_hip import synth
from oracle import az_oracle as orc
print("cpus", os.cpu_count(), [ (p["internal_api"], p["num_threads"]) for p in threadpool_info()])
head = synth.make_head(seed=1234, **synth.FULL_DIMS)
fmap = synth.make_feature_map(5, 512, 38, 63)
net = orc.OracleNet(head, feat_fn=lambda d: fmap)
nets = {"full": net, "fc": net}
cfg = orc.OracleCfg(Tz=0.0)
def run(n=2):
    orc.im_propose(nets, (600, 1000), 1.0, cfg)
    t = time.time()
    for _ in range(n): orc.im_propose(nets, (600, 1000), 1.0, cfg)
    return (time.time() - t) / n
print("numpy default: %.3f s/image" % run())
for nt in (32, 64, 128):
    with threadpool_limits(limits=nt):
        print("numpy %d threads: %.3f s/image" % (nt, run()))
for nt in (256, 128, 64, 32):
    orc.set_fc_backend("torch", threads=nt)
    print("torch %d threads (numpy pools as they are): %.3f s/image" % (nt, run()))
    with threadpool_limits(limits=1, user_api="blas"):
        print("torch %d threads, numpy BLAS 1 thread: %.3f s/image" % (nt, run()))
    x = torch.randn(517, 25088); w = torch.from_numpy(head["W6"]); b = torch.from_numpy(head["b6"])
    best = 1e9
    for _ in range(3):
        t = time.time(); torch.addmm(b, x, w.t()); best = min(best, time.time() - t)
    print("   sgemm 517x25088x4096: %.0f GFLOP/s" % (2.0 * 517 * 25088 * 4096 / best / 1e9))
orc.set_fc_backend("numpy")
