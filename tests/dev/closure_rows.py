"""closure rows per image shape (dev): AZ_FULL_DEBUG=1 python tests/dev/closure_rows.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "az-net_amd", "lib"))
from aznet_hip import ffi, synth
from aznet_hip.net import HipAZNet
net = HipAZNet(synth.make_head(seed=77, **synth.SMALL_DIMS), name="clos")
for H, W, sc in ((600, 1000, 1.0), (375, 500, 1.6), (480, 640, 1.25), (800, 1200, 0.75)):
    fh, fw = synth.conv_out_size(int(round(H * sc))), synth.conv_out_size(int(round(W * sc)))
    net.set_conv(synth.make_feature_map(1, synth.SMALL_DIMS["C"], fh, fw))
    for form in (True, "closure"):
        net.propose(ffi.AzContext.make_params(H, W, sc, 0.0, static_tree=False, full_spec=form))
        Y, st = net.propose(ffi.AzContext.make_params(H, W, sc, 0.0, static_tree=False, full_spec=form), want_stats=True)
        print(H, W, form, "form", st.search_form, "rows", list(st.pass_rows[:st.n_passes]))
