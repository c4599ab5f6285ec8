"""Longer runs of tests/test_gpu_nms_fuzz.py's adversarial cases (ties, duplicates, clusters, grid boxes, degenerate boxes,
thresholds 0 / 1 / > 1, sizes around the kernel and chunk boundaries) against the oracle's walk in the kernel's documented
tie order.  usage: nms_fuzz.py [cases] [seed]   (AZ_NMS_POLL=0: the copy-back path instead of the polled one)"""
import sys
sys.path.insert(0, "az-net_amd/lib"); sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from aznet_hip import ffi
from oracle import az_oracle as orc
import test_gpu_nms_fuzz as F
ctx = ffi.AzContext(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
bad = 0
for it in range(cases):
    dets, thresh, kind = F.make_case(rng)
    got, want = list(ctx.nms(dets, thresh)), F.reference_keep(orc, dets, thresh)
    if got != want:
        bad += 1
        print("MISMATCH case", it, "n", dets.shape[0], "kind", kind, "thresh", thresh, len(got), len(want))
print("nms fuzz: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
