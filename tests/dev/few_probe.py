"""Pruned-tree regime A/B (AZ_GEMM_FEW=0|1, lanes 1|2): the object stream (32 planted-object maps at the tuned threshold),
one image per search, ms per image; and the context's measured head-pass costs."""
import sys, time, os
sys.path.insert(0, "az-net_amd/lib"); sys.path.insert(0, ".")
import numpy as np, torch
from aznet_hip import ffi, synth
from aznet_hip.net import HipAZNet
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ohead = synth.make_object_head(seed=1234, **synth.FULL_DIMS)
net = HipAZNet(ohead, name="few", max_regions=4096)
net.ctx.set_lanes(lanes)
maps = [torch.from_numpy(synth.make_object_map(j, 512, 38, 63)).cuda().contiguous(memory_format=torch.channels_last) for j in range(32)]
net.ctx.tune_begin(len(maps) * 2 * net.ctx.max_regions)
for m in maps:
    net.set_conv(m)
    net.propose(ffi.AzContext.make_params(600, 1000, 1.0, 0.0, tune=True))
tz = net.ctx.tune_kth_largest(len(maps) * 20)[0]
net.ctx.tune_end()
prm = ffi.AzContext.make_params(600, 1000, 1.0, tz)
depth = 3
def run_set(stats=None):
    launched = 0
    for i in range(len(maps)):
        while launched < min(len(maps), i + depth):
            net.ctx.propose_launch(prm, fmap=maps[launched], producer_done=True); launched += 1
        Y, st = net.ctx.propose_fetch(want_stats=True)
        if stats is not None: stats.append(st)
run_set(); run_set()
torch.cuda.synchronize(); t0 = time.perf_counter()
sts = []
for _ in range(4): run_set(sts)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / (4 * len(maps)) * 1e3
print("lanes %d AZ_GEMM_FEW=%s: %.4f ms per image, reruns %d, passes/image %.2f, pass costs %s" % (
    lanes, os.environ.get("AZ_GEMM_FEW", "1"), ms, sum(int(s.n_reruns) for s in sts), np.mean([int(s.n_passes) for s in sts]),
    [(r, round(c)) for r, c in net.ctx.pass_costs()[:3]]))
