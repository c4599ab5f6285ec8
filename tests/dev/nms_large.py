import sys, time
sys.path.insert(0, "az-net_amd/lib"); sys.path.insert(0, ".")
import numpy as np
from aznet_hip import ffi
from oracle import az_oracle as orc
ctx = ffi.AzContext(0)
rng = np.random.RandomState(3)
for n in (20000, 50000, 100000):
    x1 = rng.uniform(0, 3000, n); y1 = rng.uniform(0, 2000, n)
    dets = np.stack([x1, y1, x1 + rng.uniform(10, 210, n), y1 + rng.uniform(10, 210, n), rng.permutation(n) / float(n)], 1).astype(np.float32)
    t0 = time.time(); got = list(ctx.nms(dets, 0.5)); t1 = time.time(); want = list(orc.nms(dets, 0.5)); t2 = time.time()
    print(n, "ok" if got == want else "MISMATCH", len(got), "gpu %.1f ms cpu %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
try:
    ctx.nms(np.zeros((600000, 5), np.float32), 0.5)
except Exception as e:
    print("600000:", e)
