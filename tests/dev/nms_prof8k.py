import sys
sys.path.insert(0, "az-net_amd/lib"); sys.path.insert(0, ".")
import numpy as np
from aznet_hip import ffi
ctx = ffi.AzContext(0)
rng = np.random.RandomState(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8129
x1 = rng.uniform(0, 900, n); y1 = rng.uniform(0, 500, n)
dets = np.stack([x1, y1, x1 + rng.uniform(10, 210, n), y1 + rng.uniform(10, 210, n), rng.permutation(n) / float(n)], 1).astype(np.float32)
for _ in range(60): ctx.nms(dets, 0.5)
