import os, sys, time, socket
sys.path.insert(0, "az-net_amd/lib"); sys.path.insert(0, ".")
import numpy as np, torch
import torch.distributed as dist
from aznet_hip import ffi, synth, dist as azdist
from aznet_hip.net import HipAZNet
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
head = synth.make_head(seed=1234, **synth.FULL_DIMS)
net = HipAZNet(head, max_regions=4096)
convs = [torch.from_numpy(synth.make_feature_map(4 + j, 512, 38, 63)).cuda().contiguous(memory_format=torch.channels_last) for j in range(4)]
p = ffi.AzContext.make_params(600, 1000, 1.0, 0.0, static_tree=False)
net.set_conv(convs[0]); net.propose(p); net.propose(p)
def loop(n, stage=False, ge=0, sync_gather=False):
    gat = azdist.DeviceGather(net.ctx, 300, max(ge, 1), dev) if (stage or ge) else None
    pend = 0; hs = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        net.ctx.propose_launch(p, fmap=convs[i % 4], producer_done=True)
        if gat is not None: gat.stage(pend)
        net.ctx.propose_fetch(want_scores=True)
        if ge:
            pend += 1
            if pend == ge:
                if sync_gather: gat.gather(pend)
                else:
                    h = gat.gather_begin(pend)
                    if hs: gat.gather_end(hs.pop(0))
                    hs.append(h)
                pend = 0
    while hs: gat.gather_end(hs.pop(0))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    print("plain %.4f | stage %.4f | async8 %.4f | async32 %.4f | sync8 %.4f | sync32 %.4f" % (
        loop(256), loop(256, True), loop(256, True, 8), loop(256, True, 32), loop(256, True, 8, True), loop(256, True, 32, True)), flush=True)
for mode in (0, 1 | 4, 2 | 4):
    net.ctx.set_profiling(0); net.ctx.set_profiling(mode)
    t = [loop(128) for _ in range(3)]
    net.ctx.last_kernel_times(); net.ctx.set_profiling(0)
    print("profiling mode", mode, ["%.4f" % x for x in t], flush=True)
# pieces
gat = azdist.DeviceGather(net.ctx, 300, 8, dev)
for name, f in [("gather_begin+end", lambda: gat.gather_end(gat.gather_begin(8))), ("gather sync", lambda: gat.gather(8))]:
    for _ in range(3): f()
    t0 = time.perf_counter()
    for _ in range(50): f()
    print(name, (time.perf_counter() - t0) / 50 * 1e3, "ms")
dist.destroy_process_group()
