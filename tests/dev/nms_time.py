import sys, time
sys.path.insert(0, "az-net_amd/lib"); sys.path.insert(0, ".")
import numpy as np
from aznet_hip import ffi
from oracle import az_oracle as orc
ctx = ffi.AzContext(0)
rng = np.random.RandomState(0)
for n in (20, 100, 256, 300, 2000):
    x1 = rng.uniform(0, 900, n); y1 = rng.uniform(0, 500, n)
    dets = np.stack([x1, y1, x1 + rng.uniform(10, 210, n), y1 + rng.uniform(10, 210, n), rng.permutation(n) / float(n)], 1).astype(np.float32)
    assert list(ctx.nms(dets, 0.5)) == list(orc.nms(dets, 0.5))
    for _ in range(50): ctx.nms(dets, 0.5)
    t0 = time.perf_counter()
    for _ in range(500): ctx.nms(dets, 0.5)
    g = (time.perf_counter() - t0) / 500 * 1e3
    t0 = time.perf_counter()
    for _ in range(50): orc.nms(dets, 0.5)
    print(n, "gpu %.4f ms   cpu oracle %.4f ms" % (g, (time.perf_counter() - t0) / 50 * 1e3))
