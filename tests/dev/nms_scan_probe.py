"""az_nms at proposal scale: keep lists against the oracle (several sizes, thresholds, seeds -- uneven and dense cases) and the
kernel times of the launch groups (HIP events on the ctx stream)."""
import sys, time
sys.path.insert(0, "az-net_amd/lib"); sys.path.insert(0, ".")
import numpy as np
from aznet_hip import ffi
from oracle import az_oracle as orc
ctx = ffi.AzContext(0)
bad = 0
for seed in range(6):
    rng = np.random.RandomState(seed)
    for n in (257, 300, 511, 640, 1000, 2000, 4097, 8129, 12000):
        for thresh, spread in ((0.5, 1.0), (0.3, 0.3), (0.7, 1.0), (0.05, 0.2)):
            x1 = rng.uniform(0, 900 * spread, n); y1 = rng.uniform(0, 500 * spread, n)
            dets = np.stack([x1, y1, x1 + rng.uniform(10, 210, n), y1 + rng.uniform(10, 210, n), rng.permutation(n) / float(n)], 1).astype(np.float32)
            got, want = list(ctx.nms(dets, thresh)), list(orc.nms(dets, thresh))
            if got != want:
                bad += 1
                print("MISMATCH", seed, n, thresh, spread, len(got), len(want))
print("mismatches:", bad)
rng = np.random.RandomState(0)
for n in (300, 2000, 8129):
    x1 = rng.uniform(0, 900, n); y1 = rng.uniform(0, 500, n)
    dets = np.stack([x1, y1, x1 + rng.uniform(10, 210, n), y1 + rng.uniform(10, 210, n), rng.permutation(n) / float(n)], 1).astype(np.float32)
    for _ in range(20): ctx.nms(dets, 0.5)
    t0 = time.perf_counter()
    for _ in range(200): ctx.nms(dets, 0.5)
    w = (time.perf_counter() - t0) / 200 * 1e3
    ctx.set_profiling(0); ctx.set_profiling(2 | 4)
    for _ in range(10): ctx.nms(dets, 0.5)
    kt = ctx.last_kernel_times(); ctx.set_profiling(0)
    by = {}
    for nm, l, ms in kt: by[nm] = by.get(nm, 0.0) + ms / 10
    print(n, "wall %.4f ms" % w, {k: round(v, 4) for k, v in by.items()}, "kept", len(ctx.nms(dets, 0.5)))
sys.exit(1 if bad else 0)
