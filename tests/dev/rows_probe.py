"""int6 / int7 launch times by row count through az_head_forward (HIP events per launch group): AZ_GEMM_QUART=0|1 A/B."""
import sys
sys.path.insert(0, "az-net_amd/lib"); sys.path.insert(0, ".")
import numpy as np, torch
from aznet_hip import ffi, synth
from aznet_hip.net import HipAZNet
head = synth.make_head(seed=1234, **synth.FULL_DIMS)
net = HipAZNet(head, name="rows", max_regions=4096)
net.set_conv(synth.make_feature_map(3, 512, 38, 63))
rng = np.random.RandomState(1)
for rows in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "32,33,36,40,41,48,64,72,104".split(","))]:
    x1 = rng.uniform(0, 900, rows); y1 = rng.uniform(0, 500, rows)
    rr = np.stack([np.zeros(rows), x1, y1, x1 + rng.uniform(16, 300, rows), y1 + rng.uniform(16, 300, rows)], 1).astype(np.float32)
    for _ in range(5): net.ctx.head_forward(rr)
    net.ctx.set_profiling(0); net.ctx.set_profiling(2 | 4)
    for _ in range(10): net.ctx.head_forward(rr)
    kt = net.ctx.last_kernel_times(); net.ctx.set_profiling(0)
    by = {}
    for nm, l, ms in kt: by.setdefault(nm, []).append(ms * 1e3)
    print(rows, {k: round(float(np.mean(v)), 1) for k, v in by.items()})
