#!/usr/bin/env python3
"""Builder's probe: the fp32 VGG16 conv1_1..conv5_3 forward (aznet_hip.backbone.VGG16Conv5, MIOpen convolutions + this
library's fused epilogues) at batch 1 / 2 / 4 / 8 of 600x1000 blobs: ms per image, and whether a batched map equals the
batch-1 map of the same image bit for bit."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "az-net_amd", "lib"))


def main():
    import torch
    from aznet_hip.backbone import VGG16Conv5
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda", 0)
    bb = VGG16Conv5(device=dev, seed=4321, channels_last_out=True, channels_last_compute=True)
    x8 = torch.randn(8, 3, 600, 1000, device=dev).contiguous(memory_format=torch.channels_last)
    ref = [bb(x8[i:i + 1]).clone() for i in range(8)]
    for bs in (1, 2, 4, 8):
        xs = [x8[i:i + bs] for i in range(0, 8, bs)]
        for _ in range(3):
            for x in xs:
                y = bb(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            for x in xs:
                y = bb(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 / 8 * 1e3
        same = all(torch.equal(bb(x8[i:i + bs])[0], ref[i][0]) for i in range(0, 8, bs))
        print("batch %d: %.3f ms per image; first map of each batch equals its batch-1 map bit for bit: %s" % (bs, ms, same))


if __name__ == "__main__":
    main()
