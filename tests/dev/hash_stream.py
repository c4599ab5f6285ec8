#!/usr/bin/env python3
"""Builder's probe: hash of every result of the object stream (hot path only, fixed maps) -- solo vs sharing the GPU."""
import os, sys, hashlib
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "az-net_amd", "lib")); sys.path.insert(0, REPO)
import torch
from aznet_hip import ffi, synth
from aznet_hip.net import HipAZNet
from aznet_hip.backbone import VGG16Conv5
mode = sys.argv[1] if len(sys.argv) > 1 else "hot"
H, W = 600, 1000
h = hashlib.sha256()
if mode == "hot":
    head = synth.make_object_head(seed=1234, **synth.FULL_DIMS)
    net = HipAZNet(head, name="p", max_regions=4096)
    net.ctx.set_lanes(2)
    maps = [torch.from_numpy(synth.make_object_map(j, 512, 38, 63)).cuda().contiguous(memory_format=torch.channels_last) for j in range(32)]
    prm = ffi.AzContext.make_params(H, W, 1.0, 0.51)
    seq = list(range(32)) * 6
    launched = 0
    for i in range(len(seq)):
        while launched < min(len(seq), i + 3):
            net.ctx.propose_launch(prm, fmap=maps[seq[launched]], producer_done=True); launched += 1
        Y, S = net.ctx.propose_fetch(want_scores=True)
        h.update(Y.tobytes()); h.update(S.tobytes())
else:
    # backbone only: conv5_3 of 7 noise images, deterministic flag as given by the environment
    if os.environ.get("AZ_BACKBONE_DETERMINISTIC", "0") != "0":
        torch.backends.cudnn.deterministic = True
    bb = VGG16Conv5(device="cuda:0", seed=1235)
    for rep in range(3):
        for j in range(7):
            im = synth.make_image(j, H, W).astype(np.float32).transpose(2, 0, 1)[None] - 110.0
            c = bb(im)
            h.update(c.cpu().numpy().tobytes())
print(mode, h.hexdigest()[:16])
