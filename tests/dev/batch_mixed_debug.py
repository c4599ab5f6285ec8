#!/usr/bin/env python3
"""Builder's probe: a lockstep batch whose images differ in shape and in level count; per-image form / reruns / levels."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "az-net_amd", "lib"))
import torch
from aznet_hip import ffi, synth
from aznet_hip.net import HipAZNet
head = synth.make_head(seed=77, **synth.SMALL_DIMS)
net = HipAZNet(head, name="dbg")
shapes = [(600, 1000), (160, 240), (375, 500), (90, 130), (1000, 700), (500, 375)]
prm, tm = [], []
for j, (H, W) in enumerate(shapes):
    sc = 600.0 / min(H, W)
    if round(sc * max(H, W)) > 1000:
        sc = 1000.0 / max(H, W)
    fh, fw = synth.conv_out_size(int(round(H * sc))), synth.conv_out_size(int(round(W * sc)))
    tm.append(torch.from_numpy(synth.make_scene_map(400 + j, synth.SMALL_DIMS["C"], fh, fw)).cuda().contiguous(memory_format=torch.channels_last))
    prm.append(ffi.AzContext.make_params(H, W, sc, float(sys.argv[1]) if len(sys.argv) > 1 else 0.0, static_tree=False))
for rep in range(2):
    net.ctx.batch_launch(prm, tm, producer_done=True)
    for (Y, st), sh in zip(net.ctx.batch_fetch_all(want_stats=True), shapes):
        print(rep, sh, "form", st.search_form, "reruns", st.n_reruns, "levels", st.n_levels, [int(st.level_regions[l]) for l in range(st.n_levels)],
              "passes", [int(x) for x in list(st.pass_rows)[:st.n_passes]])
