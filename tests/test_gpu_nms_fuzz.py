"""az_nms (lib/utils/nms.pyx:17-68) against the oracle's greedy walk on adversarial inputs: quantised scores (ties),
duplicated boxes, tight clusters (long suppression chains inside a 64-box chunk), integer-grid boxes (IoU on simple
fractions, `ovr == thresh` included), degenerate boxes (x2 < x1), thresholds 0 / 1 / > 1, sizes around the small / large
kernel boundary (256) and the chunk boundaries.  The order among EQUAL scores is the kernel's documented one -- the higher
original index first, what a stable ascending sort reversed yields; numpy's default argsort (nms.pyx:25) leaves it to its
quicksort, so the reference's own order of ties is not defined -- and the oracle's C walk is given that order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 512, 513, 1000, 2049, 4097]


def make_case(rng):
    n = int(rng.choice(SIZES + [int(rng.randint(1, 6000))]))
    kind = int(rng.randint(5))
    if kind == 0:      # uniform
        x1 = rng.uniform(0, 900, n); y1 = rng.uniform(0, 500, n); w = rng.uniform(10, 210, n); h = rng.uniform(10, 210, n)
    elif kind == 1:    # a few tight clusters: long chains
        k = max(1, n // 50); cx = rng.uniform(0, 900, k); cy = rng.uniform(0, 500, k); a = rng.randint(k, size=n)
        x1 = cx[a] + rng.normal(0, 6, n); y1 = cy[a] + rng.normal(0, 6, n); w = 80 + rng.normal(0, 5, n); h = 60 + rng.normal(0, 5, n)
    elif kind == 2:    # many exact duplicates
        m = max(1, n // 4); bx = rng.uniform(0, 900, m); by = rng.uniform(0, 500, m); a = rng.randint(m, size=n)
        x1 = bx[a]; y1 = by[a]; w = np.full(n, 50.0); h = np.full(n, 40.0)
    elif kind == 3:    # integer grid boxes (IoU exactly on simple fractions)
        x1 = rng.randint(0, 40, n) * 8.0; y1 = rng.randint(0, 30, n) * 8.0
        w = rng.randint(1, 6, n) * 8.0 - 1; h = rng.randint(1, 6, n) * 8.0 - 1
    else:              # degenerate boxes mixed in
        x1 = rng.uniform(0, 900, n); y1 = rng.uniform(0, 500, n); w = rng.uniform(-20, 100, n); h = rng.uniform(-20, 100, n)
    sc = rng.permutation(n) / float(n) if rng.rand() < 0.5 else np.round(rng.uniform(0, 1, n) * 16) / 16.0
    dets = np.stack([x1, y1, x1 + w, y1 + h, sc], 1).astype(np.float32)
    thresh = float(rng.choice([0.3, 0.5, 0.7, 0.0, 1.0, 1.5, 0.25, 1.0 / 3.0, rng.uniform(0, 1)]))
    return dets, thresh, kind


def reference_keep(orc, dets, thresh):
    n = dets.shape[0]
    order = np.ascontiguousarray(np.lexsort((np.arange(n), dets[:, 4]))[::-1], dtype=np.int64)
    keep = np.zeros(n, dtype=np.int64)
    nk = orc.lib().orc_nms(orc._fp(dets), n, orc._lp(order), float(thresh), orc._lp(keep))
    return [int(k) for k in keep[:nk]]


def test_nms_adversarial_inputs_match_the_oracle_walk():
    from aznet_hip import ffi
    from oracle import az_oracle as orc
    ctx = ffi.AzContext(0)
    rng = np.random.RandomState(2024)
    for it in range(250):
        dets, thresh, kind = make_case(rng)
        got = list(ctx.nms(dets, thresh))
        assert got == reference_keep(orc, dets, thresh), (it, dets.shape[0], kind, thresh)
