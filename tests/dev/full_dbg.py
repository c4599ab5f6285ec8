import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "az-net_amd", "lib")); sys.path.insert(0, os.path.join(HERE, "..", ".."))
from aznet_hip import ffi, synth
from aznet_hip.net import HipAZNet
head = synth.make_head(seed=77, **synth.SMALL_DIMS)
net = HipAZNet(head, name="dbg")
net.set_conv(synth.make_feature_map(5, synth.SMALL_DIMS["C"], 38, 63))
for Tz in (0.0, 0.5):
    for full in (False, True, True):
        Y, S, st = net.propose(ffi.AzContext.make_params(600, 1000, 1.0, Tz, static_tree=False, full_spec=full), want_scores=True, want_stats=True)
        print("Tz", Tz, "full", full, "passes", st.n_passes, list(st.pass_rows[:4]), "U", list(st.level_unique[:6]), "sum", float(S.sum()))
