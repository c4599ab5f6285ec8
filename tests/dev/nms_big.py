"""az_nms at proposal-scale N against the NumPy restatement + kernel time (dev script): python tests/dev/nms_big.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "az-net_amd", "lib"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__)))
from aznet_hip import ffi
from nms_flaky import ref_nms
ctx = ffi.AzContext(0)
rng = np.random.RandomState(0)
for n in (257, 300, 2000, 8129, 8193, 20000):
    x1 = rng.uniform(0, 900, n); y1 = rng.uniform(0, 500, n)
    dets = np.stack([x1, y1, x1 + rng.uniform(10, 210, n), y1 + rng.uniform(10, 210, n), rng.permutation(n) / float(n)], 1).astype(np.float32)
    k = ctx.nms(dets, 0.5)
    ok = (list(k) == ref_nms(dets, 0.5)) if n <= 8200 else None
    for _ in range(3): ctx.nms(dets, 0.5)
    t = time.perf_counter()
    for _ in range(10): ctx.nms(dets, 0.5)
    wall = (time.perf_counter() - t) / 10 * 1e3
    ctx.set_profiling(2 | 4)
    for _ in range(5): ctx.nms(dets, 0.5)
    kt = ctx.last_kernel_times(); ctx.set_profiling(0)
    print("n %6d kept %5d ok %s wall %.3f ms kernel %.3f ms" % (n, len(k), ok, wall, sum(ms for _, _, ms in kt) / 5))
