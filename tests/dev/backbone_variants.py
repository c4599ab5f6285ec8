"""Timing of the PyTorch-ROCm VGG16 conv5_3 forward (plumbing, not the product) under a few library settings."""
import sys, time
sys.path.insert(0, "az-net_amd/lib"); sys.path.insert(0, ".")
import torch
from aznet_hip.backbone import VGG16Conv5

def bench(bb, x, n=20):
    for _ in range(5): bb(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): bb(x)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

x = torch.randn(1, 3, 600, 1000, device="cuda")
bb = VGG16Conv5(device="cuda:0", seed=1)
print("default                 %.3f ms" % bench(bb, x), flush=True)
torch.backends.cudnn.benchmark = True
print("cudnn.benchmark         %.3f ms" % bench(bb, x), flush=True)
xc = x.contiguous(memory_format=torch.channels_last)
bb2 = VGG16Conv5(device="cuda:0", seed=1)
bb2.layers = [None if l is None else (l[0], l[1].contiguous(memory_format=torch.channels_last), l[2]) for l in bb2.layers]
print("channels_last + bench   %.3f ms" % bench(bb2, xc), flush=True)
torch.backends.cudnn.benchmark = False
print("channels_last           %.3f ms" % bench(bb2, xc), flush=True)
y1 = bb(x); y2 = bb2(xc)
print("max diff", float((y1 - y2).abs().max()), float(y1.abs().max()))
try:
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    print("allow_tf32 off          %.3f ms" % bench(bb, x), flush=True)
except Exception as e:
    print(e)
