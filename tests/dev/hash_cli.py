#!/usr/bin/env python3
"""Builder's probe: the CLI's queue-ahead loop (detect.test._propose_start / _propose_finish) on 7 synthetic images, a hash per
image of the blob, of conv5_3 and of the proposals -- solo vs sharing the GPU."""
import os, sys, hashlib, io
from contextlib import redirect_stdout
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "az-net_amd", "lib")); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "az-net_amd", "tools"))
import torch
import prop_az
from detect import config as C
from detect import test as T
from datasets.factory import get_imdb
C.cfg_set_mode("Test", 0.35)
net = prop_az.load_net("synthetic", 0)
if os.environ.get("DBG_NOFUSE"): net.backbone.fused_epilogue = False
EARLY = bool(os.environ.get("DBG_EARLY"))
imdb = get_imdb("synthetic_600x1000_7")
nets = {"full": net, "fc": net}
def hh(a): return hashlib.sha256(a.tobytes()).hexdigest()[:8]
rows = []
pend = None
with redirect_stdout(io.StringIO()):
    for rep in range(2):
        for i in range(8):
            nxt = T._propose_start(nets, imdb.image_at(i), after=(pend["done"] if pend is not None else None)) if i < 7 else None
            if nxt is not None and EARLY:
                nxt["conv_early"] = hh(nxt["conv"].cpu().numpy())
            if pend is not None:
                Y = T._propose_finish(nets, pend)
                rows.append((hh(pend["blob"].cpu().numpy()), pend.get("conv_early", "-"), hh(pend["conv"].cpu().numpy()), Y.shape[0], hh(Y)))
            pend = nxt
for r in rows: print(*r)
