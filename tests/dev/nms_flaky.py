"""Stress of az_nms / az_nms_batched against a NumPy restatement of nms.pyx with this library's tie rule
(descending score, HIGHER index first): which path, if any, ever returns a wrong keep list?
    python tests/dev/nms_flaky.py [rounds]         (AZ_NMS_POLL=0 for the stream-wait form)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "az-net_amd", "lib"))
from aznet_hip import ffi


def ref_nms(d, thresh):
    n = d.shape[0]
    if n == 0:
        return []
    x1, y1, x2, y2, sc = [d[:, i] for i in range(5)]
    areas = (x2 - x1 + np.float32(1)) * (y2 - y1 + np.float32(1))
    order = np.argsort(sc, kind="stable")[::-1]
    sup = np.zeros(n, dtype=bool)
    keep = []
    for a in range(n):
        i = order[a]
        if sup[i]:
            continue
        keep.append(int(i))
        rest = order[a + 1:]
        xx1 = np.maximum(x1[i], x1[rest]); yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest]); yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1 + np.float32(1)); h = np.maximum(np.float32(0), yy2 - yy1 + np.float32(1))
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        sup[rest[ovr.astype(np.float64) >= thresh]] = True
    return keep


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    ctx = ffi.AzContext(0)
    rng = np.random.RandomState(17)
    sets = []
    for n in [0, 1, 2, 63, 64, 65, 100, 255, 256, 257, 300, 1000] + [int(v) for v in rng.randint(1, 200, 80)]:
        x1 = rng.uniform(0, 300, n); y1 = rng.uniform(0, 300, n)
        d = np.stack([x1, y1, x1 + rng.uniform(5, 150, n), y1 + rng.uniform(5, 150, n), rng.uniform(0, 1, n)], 1).astype(np.float32)
        if n > 4:
            d[3] = d[1]
            d[4, :4] = d[0, :4]
        sets.append(d)
    refs = {t: [ref_nms(d, t) for d in sets] for t in (0.3, 0.5)}
    bad = {"batched": 0, "single": 0}
    detail = []
    for r in range(rounds):
        for t in (0.3, 0.5):
            got = ctx.nms_batched(sets, t)
            for gi, (d, k) in enumerate(zip(sets, got)):
                if list(k) != refs[t][gi]:
                    bad["batched"] += 1
                    detail.append(("batched", r, t, gi, len(d), len(k), len(refs[t][gi])))
                k1 = ctx.nms(d, t)
                if list(k1) != refs[t][gi]:
                    bad["single"] += 1
                    detail.append(("single", r, t, gi, len(d), len(k1), len(refs[t][gi])))
    print("rounds", rounds, "AZ_NMS_POLL", os.environ.get("AZ_NMS_POLL"), "mismatches", bad)
    for x in detail[:40]:
        print("  ", x)


if __name__ == "__main__":
    main()
