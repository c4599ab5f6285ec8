"""GPU parity tests: every stage of the HIP path, called through the C ABI
(aznet_hip.ffi -> libaznet_hip.so), against the oracle and the golden vectors.

Tolerances: integer / index / f64-geometry work is compared bit-exactly.  The fp32 head
is compared at 1e-4 (north_star); decoded boxes at 1e-4 px absolute, because the only inexact
operation of the decode is the f32 exp (NumPy's SIMD expf vs the device's differ by an ulp,
which a 1000-px box width turns into <= 4e-5 px).
"""
import numpy as np
import pytest

from helpers import load, unpack_list, TRACES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from aznet_hip import ffi, synth
    from aznet_hip.net import HipAZNet
    from oracle import az_oracle as orc
    return ffi, synth, HipAZNet, orc


@pytest.fixture(scope="module")
def small(mods):
    ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    net = HipAZNet(head, name="small")
    return net, head


@pytest.fixture(scope="module")
def full(mods):
    ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=1234, **synth.FULL_DIMS)
    net = HipAZNet(head, name="full", max_regions=4096)
    return net, head


# ---------------------------------------------------------------- geometry, bit-exact
def test_divide_region_golden(small):
    ctx = small[0].ctx
    g = load("g1_divide_region.npz")
    for i in range(len(g["sizes"])):
        for a, b in zip(unpack_list(g, "root%d_in" % i), unpack_list(g, "root%d_out" % i)):
            got = ctx.divide_region(a, 10.0)
            assert got.shape == b.shape and np.array_equal(got, b)
    assert np.array_equal(ctx.divide_region(g["rand_in"], 10.0), g["rand_out"])
    for i, s in enumerate(unpack_list(g, "single_out")):
        assert np.array_equal(ctx.divide_region(g["rand_in"][i:i + 1], 10.0), s)
    assert ctx.divide_region(np.zeros((0, 4)), 10.0).shape == (0, 4)


def test_sift_dup_golden(small):
    ctx = small[0].ctx
    g = load("g2_sift_dup.npz")
    assert np.array_equal(ctx.sift_dup(g["in"], 10.0), g["out10"])
    assert np.array_equal(ctx.sift_dup(g["in"], 16.0), g["out16"])


def test_cython_dropins(small, mods):
    import utils.cython_div as cdiv
    import utils.cython_nms as cnms
    g = load("g1_divide_region.npz")
    assert np.array_equal(cdiv.divide_region(g["rand_in"], 10.0), g["rand_out"])
    with pytest.raises(ValueError):
        cdiv.divide_region(g["rand_in"].astype(np.float32), 10.0)
    with pytest.raises(ValueError):
        cnms.nms(np.zeros((3, 5)), 0.5)
    assert cnms.nms(np.zeros((0, 5), dtype=np.float32), 0.5) == []


@pytest.mark.parametrize("scale,batch", [(1.0, 10000), (1.6, 10000), (0.9375, 100), (1.0, 7)])
def test_roi_dedup_vs_oracle(small, mods, scale, batch):
    ffi, synth, HipAZNet, orc = mods
    ctx = small[0].ctx
    B = np.array([[0, 0, 999.0, 599.0]])
    for _ in range(5):
        rois, index, inv = ctx.roi_dedup(B, scale, 1. / 16., batch)
        ref_rois = orc.get_rois_blob(B, scale)
        assert np.array_equal(rois, ref_rois)
        # oracle dedup is per BATCH_SIZE chunk (test.py:195-218)
        ref_index, ref_inv, off = [], [], 0
        for s in range(0, B.shape[0], batch):
            idx, iv = orc.roi_dedup(ref_rois[s:s + batch])
            ref_index.append(idx + s)
            ref_inv.append(iv + off)
            off += len(idx)
        assert np.array_equal(index, np.concatenate(ref_index))
        assert np.array_equal(inv, np.concatenate(ref_inv))
        B = orc.divide_region(B, 10)


def test_decode_filter_golden(small):
    ctx = small[0].ctx
    g = load("g4_decode.npz")
    b, s = ctx.decode_filter(g["boxes"], g["deltas"], g["scores"], 600, 1000)
    assert b.shape == g["unwrap_boxes"].shape            # same candidates survive the filter
    np.testing.assert_allclose(b, g["unwrap_boxes"], rtol=1e-6, atol=1e-4)   # px; f32-exp ulps x box size
    assert np.array_equal(s, g["unwrap_scores"])


def test_nms_golden(small):
    ctx = small[0].ctx
    g = load("g5_nms.npz")
    for i in range(int(g["ncases"])):
        keep = ctx.nms(g["dets%d" % i], float(g["thresh%d" % i]))
        assert list(keep) == list(g["keep%d" % i]), "nms case %d" % i


def test_nms_batched_equals_single_calls(small, mods):
    """az_nms_batched (one workgroup per small group, LDS-resident) == az_nms per group == oracle."""
    ffi, synth, HipAZNet, orc = mods
    ctx = small[0].ctx
    rng = np.random.RandomState(17)
    sets = []
    for n in [0, 1, 2, 63, 64, 65, 100, 255, 256, 257, 300, 1000] + [int(v) for v in rng.randint(1, 200, 80)]:
        x1 = rng.uniform(0, 300, n); y1 = rng.uniform(0, 300, n)
        d = np.stack([x1, y1, x1 + rng.uniform(5, 150, n), y1 + rng.uniform(5, 150, n), rng.uniform(0, 1, n)], 1)
        d = d.astype(np.float32)
        if n > 4:
            d[3] = d[1]                                   # duplicate box, equal score (tie order)
            d[4, :4] = d[0, :4]
        sets.append(d)
    for thresh in (0.3, 0.5):
        got = ctx.nms_batched(sets, thresh)
        for d, k in zip(sets, got):
            assert list(k) == list(ctx.nms(d, thresh))
            # (equal scores: NumPy's argsort order is unspecified, so the oracle only judges the tie-free sets)
            u = d.copy()
            u[:, 4] = (np.argsort(np.argsort(d[:, 4], kind="stable"), kind="stable") + 1) / float(max(len(d), 1))
            assert list(ctx.nms_batched([u], thresh)[0]) == list(orc.nms(u, thresh))
    assert ctx.nms_batched([], 0.5) == []


def test_topk_vs_numpy(small):
    ctx = small[0].ctx
    rng = np.random.RandomState(5)
    for n, k in [(1, 300), (299, 300), (300, 300), (301, 300), (8129, 300), (30019, 300), (5000, 2000)]:
        s = rng.uniform(0, 1, n).astype(np.float32)
        if n > 1000:
            s[rng.randint(0, n, n // 4)] = s[rng.randint(0, n, n // 4)]      # ties
        idx = ctx.topk(s, k)
        ref = np.argsort(-s.astype(np.float64), kind="stable")[:k]
        assert np.array_equal(idx, ref), (n, k)


def test_roi_dedup_golden_g3(small):
    """az_roi_dedup against index / inv_index recorded from the reference's own np.unique inside _az_forward
    (test.py:212-218; tests/golden/g3_roi_dedup.npz: scales 1.0 / 1.6 / 0.9375, chunked levels)."""
    ctx = small[0].ctx
    g = load("g3_roi_dedup.npz")
    by_level = {}
    for i in range(int(g["ncases"])):
        boxes, scale = g["c%d_boxes" % i], float(g["c%d_scale" % i])
        rois, index, inv = ctx.roi_dedup(boxes, scale, 1. / 16., 10000)
        assert np.array_equal(index, g["c%d_index" % i]) and np.array_equal(inv, g["c%d_inv_index" % i]), i
        v = np.array([1, 1e3, 1e6, 1e9, 1e12])
        assert np.array_equal(np.round(rois * np.float32(1. / 16.)).dot(v), g["c%d_hashes" % i])
        by_level.setdefault((int(g["c%d_H" % i]), int(g["c%d_W" % i]), int(g["c%d_batch" % i])), []).append(i)
    # the chunked cases once more as ONE call with BATCH_SIZE 100: the library chunks as test.py:195-205 does
    for (H, W, batch), ids in by_level.items():
        if batch != 100:
            continue
        big = [i for i in ids if g["c%d_boxes" % i].shape[0] == 100 or i == ids[-1]][-4:]
        boxes = np.vstack([g["c%d_boxes" % i] for i in big])
        rois, index, inv = ctx.roi_dedup(boxes, float(g["c%d_scale" % big[0]]), 1. / 16., 100)
        ref_index, ref_inv, off, s = [], [], 0, 0
        for i in big:
            ref_index.append(g["c%d_index" % i] + s)
            ref_inv.append(g["c%d_inv_index" % i] + off)
            off += g["c%d_index" % i].shape[0]
            s += g["c%d_boxes" % i].shape[0]
        assert np.array_equal(index, np.concatenate(ref_index)) and np.array_equal(inv, np.concatenate(ref_inv))


def test_topk_golden_g8(small):
    """az_topk against the reference's own `np.argsort(-aScores)` of whole im_propose runs (test.py:397-401;
    tests/golden/g8_topk.npz).  NumPy's order inside a tie is unspecified: the score sequence must agree, the index set
    above the last selected score must agree, and where ties are duplicates of one RoI (identical boxes) the selected BOXES
    must agree."""
    ctx = small[0].ctx
    g = load("g8_topk.npz")
    for tag in [str(t) for t in g["runs"]]:
        neg, indA, Yall, Y = g[tag + "_neg_scores"], g[tag + "_indA"], g[tag + "_Y_all"], g[tag + "_Y"]
        k = int(g[tag + "_num_proposals"])
        sc = (-neg).astype(np.float32)
        assert np.array_equal(sc.astype(np.float64), -neg)               # (aScores hold f32 values, test.py:381)
        idx = ctx.topk(sc, k)
        n = min(k, sc.shape[0])
        assert idx.shape == (n,)
        assert np.array_equal(neg[idx], neg[indA[:n]]), tag
        cut = neg[indA[n - 1]]
        assert set(idx[neg[idx] < cut].tolist()) == set(indA[:n][neg[indA[:n]] < cut].tolist()), tag
        # the boxes the reference returned, tie groups compared as sets (equal scores: the same RoIPool window met from
        # different regions -- same score, boxes decoded against different anchors)
        for v in np.unique(neg[indA[:n]]):
            if v == cut:
                continue
            a = Yall[idx[neg[idx] == v]]
            b = Y[neg[indA[:n]] == v]
            assert np.array_equal(a[np.lexsort(a.T[::-1])], b[np.lexsort(b.T[::-1])]), tag


# ---------------------------------------------------------------- head
def _rand_rois(rng, n, W, H):
    x1 = rng.uniform(0, W - 20, n)
    y1 = rng.uniform(0, H - 20, n)
    w = rng.uniform(8, W / 2, n)
    h = rng.uniform(8, H / 2, n)
    r = np.stack([np.zeros(n), x1, y1, np.minimum(x1 + w, W - 1), np.minimum(y1 + h, H - 1)], 1)
    r[0] = [0, 0, 0, W - 1, H - 1]
    if n > 3:
        r[1] = [0, 8, 8, 8, 8]                       # .5 rounding, one cell
        r[2] = [0, W - 5, H - 5, W + 200, H + 300]   # hangs off the map -> empty bins
    return r.astype(np.float32)


def test_roi_pool_bit_exact(small, mods):
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    fmap = synth.make_feature_map(3, synth.SMALL_DIMS["C"], 38, 63)
    net.set_conv(fmap)
    rois = _rand_rois(np.random.RandomState(1), 300, 1000, 600)
    got = net.ctx.roi_pool(rois)
    ref = orc.roi_pool(fmap[0], rois)
    assert np.array_equal(got, ref)


def test_roi_pool_full_width_bit_exact(full, mods):
    ffi, synth, HipAZNet, orc = mods
    net, head = full
    fmap = synth.make_feature_map(4, 512, 38, 63)
    net.set_conv(fmap)
    rois = _rand_rois(np.random.RandomState(2), 40, 1000, 600)
    assert np.array_equal(net.ctx.roi_pool(rois), orc.roi_pool(fmap[0], rois))


@pytest.mark.parametrize("R", [1, 8, 41, 130, 300])
def test_head_forward_small(small, mods, R):
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    fmap = synth.make_feature_map(3, synth.SMALL_DIMS["C"], 38, 63)
    net.set_conv(fmap)
    rois = _rand_rois(np.random.RandomState(R), R, 1000, 600)
    z, p, d = net.ctx.head_forward(rois)
    zr, pr, dr = orc.head_forward(head, fmap[0], rois)
    np.testing.assert_allclose(z, zr, rtol=0, atol=1e-4)
    np.testing.assert_allclose(p, pr, rtol=0, atol=1e-4)
    np.testing.assert_allclose(d, dr, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("R", [1, 33, 130])
def test_head_forward_full(full, mods, R):
    """Full-size head (25088 -> 4096 -> {1024 -> 11+44, 256 -> 1}) vs the BLAS oracle."""
    ffi, synth, HipAZNet, orc = mods
    net, head = full
    fmap = synth.make_feature_map(4, 512, 38, 63)
    net.set_conv(fmap)
    rois = _rand_rois(np.random.RandomState(R), R, 1000, 600)
    z, p, d = net.ctx.head_forward(rois)
    zr, pr, dr = orc.head_forward(head, fmap[0], rois)
    np.testing.assert_allclose(z, zr, rtol=0, atol=1e-4)
    np.testing.assert_allclose(p, pr, rtol=0, atol=1e-4)
    np.testing.assert_allclose(d, dr, rtol=1e-4, atol=1e-4)


def test_head_rows_independent_of_batch(small, mods):
    """Each roi's outputs are a fixed function of that roi: same bits whatever else is in
    the batch (needed for chunk-independence and for unit-vs-fused agreement)."""
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    fmap = synth.make_feature_map(3, synth.SMALL_DIMS["C"], 38, 63)
    net.set_conv(fmap)
    rois = _rand_rois(np.random.RandomState(9), 200, 1000, 600)
    z, p, d = net.ctx.head_forward(rois)
    z1, p1, d1 = net.ctx.head_forward(rois[37:38])
    z2, p2, d2 = net.ctx.head_forward(rois[100:170])
    assert np.array_equal(z[37:38], z1) and np.array_equal(p[37:38], p1) and np.array_equal(d[37:38], d1)
    assert np.array_equal(z[100:170], z2) and np.array_equal(p[100:170], p2) and np.array_equal(d[100:170], d2)


@pytest.mark.parametrize("which", ["small", "full"])
def test_head_rows_same_bits_in_full_half_and_padded_strips(small, full, mods, which):
    """A roi's outputs must not depend on where the launch's row count puts it: a full 32-row strip
    (32x32x2 MFMA), the 16-row half strip at the end (16x16x4 MFMA) or a padded last strip."""
    ffi, synth, HipAZNet, orc = mods
    net, head = small if which == "small" else full
    C = synth.SMALL_DIMS["C"] if which == "small" else 512
    net.set_conv(synth.make_feature_map(3, C, 38, 63))
    rois = _rand_rois(np.random.RandomState(10), 300, 1000, 600)
    ref = net.ctx.head_forward(rois)                       # 300 rows = 9 strips + 12 rows (half strip)
    sizes = list(range(1, 70)) + [95, 96, 97, 112, 113, 128, 129, 130, 144, 145, 160, 161, 177, 193, 256, 257, 288, 289]
    if which == "full":
        sizes = [1, 31, 33, 40, 48, 49, 64, 65, 80, 81, 96, 100, 130, 145, 161, 257, 289]
    for n in sizes:
        got = net.ctx.head_forward(rois[:n])
        for a, b in zip(got, ref):
            assert np.array_equal(a, b[:n]), n
        lo = max(0, 300 - n)
        got = net.ctx.head_forward(rois[lo:])              # the same rois at other row positions
        for a, b in zip(got, ref):
            assert np.array_equal(a, b[lo:]), n


def test_many_row_gemm_same_bits_as_the_tile_gemm(full, mods):
    """Launches whose row count the host knows (az_head_forward, the one-pass search) take the 12-wave many-row
    GEMM (az_head12.hip) from 161 rows on: m-tiles of <= 12 strip slots dealt to three row groups, the trailing
    <= 16 rows as a half strip.  A roi's bits must not depend on the kernel, the m-tile, the row group or the
    kind of strip it lands in: sub-batches (<= 160 rows: k_fc_splitk) and shifted windows against a 1000-row launch."""
    ffi, synth, HipAZNet, orc = mods
    net, head = full
    net.set_conv(synth.make_feature_map(3, 512, 38, 63))
    rois = _rand_rois(np.random.RandomState(11), 1000, 1000, 600)
    ref = net.ctx.head_forward(rois)                       # 1000 rows = 31 strips + 8 rows: 3 m-tiles, half strip
    for n in [64, 128, 160, 161, 176, 177, 192, 193, 200, 224, 225, 256, 257, 272, 273, 288, 300, 352, 353, 368, 369, 384, 385, 400, 401, 517, 688, 689, 700, 704,
              705, 720, 721, 737, 768, 769, 784, 785, 999]:
        got = net.ctx.head_forward(rois[:n])
        for a, b in zip(got, ref):
            assert np.array_equal(a, b[:n]), n
        lo = 1000 - n
        got = net.ctx.head_forward(rois[lo:])              # the same rois at other row positions
        for a, b in zip(got, ref):
            assert np.array_equal(a, b[lo:]), n


def test_many_row_gemm_is_right_for_any_row_count(full, mods, monkeypatch):
    """In the level loop only the device knows a level's row count; a level that forwarded many rois in the previous
    search is sent to the many-row GEMM whatever it holds this time.  That kernel must therefore be right -- and give
    the tile GEMM's bits -- for ANY row count, down to one row (AZ_GEMM12_MIN=1 sends every launch there)."""
    ffi, synth, HipAZNet, orc = mods
    net, head = full
    fmap = synth.make_feature_map(3, 512, 38, 63)
    net.set_conv(fmap)
    rois = _rand_rois(np.random.RandomState(12), 200, 1000, 600)
    ref = net.ctx.head_forward(rois)                        # (200 rows: many-row kernel by default; checked above)
    small = {n: net.ctx.head_forward(rois[:n]) for n in (1, 5, 16, 17, 31, 32, 33, 48, 49, 64, 100, 129, 160)}   # tile GEMM
    monkeypatch.setenv("AZ_GEMM12_MIN", "1")
    net12 = HipAZNet(head, name="all12", max_regions=4096)
    net12.set_conv(fmap)
    for n, want in small.items():
        got = net12.ctx.head_forward(rois[:n])
        for a, b, r in zip(got, want, ref):
            assert np.array_equal(a, b), n
            assert np.array_equal(a, r[:n]), n


# ---------------------------------------------------------------- whole loop
def _oracle_loop_on_gpu_head(orc, net, fmap, H, W, scale, cfg):
    """The oracle's level loop with the HIP head injected as the pycaffe-shaped net --
    the same seam the reference's Python uses (test.py:221-236)."""
    return orc.im_propose({"full": net, "fc": net}, (H, W), scale, cfg, data_blob=None,
                          return_trace=True)


@pytest.mark.parametrize("H,W,tzq,batch", [(600, 1000, 0.0, 10000), (375, 500, 0.55, 10000),
                                           (480, 640, 0.4, 10000), (640, 853, 0.5, 100),
                                           (600, 1000, 1.5, 10000),
                                           (800, 1200, 0.0, 10000)])        # BASELINE config 4: deep tree, K = 7
def test_fused_loop_equals_per_level_loop(small, mods, H, W, tzq, batch):
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    scale = 600.0 / min(H, W)
    if np.round(scale * max(H, W)) > 1000:
        scale = 1000.0 / max(H, W)
    fh, fw = synth.conv_out_size(int(round(H * scale))), synth.conv_out_size(int(round(W * scale)))
    fmap = synth.make_feature_map(5, synth.SMALL_DIMS["C"], fh, fw)
    net.set_conv(fmap)
    # choose Tz from the zoom scores of a full-tree run
    if tzq in (0.0, 1.5):
        Tz = tzq
    else:
        p0 = ffi.AzContext.make_params(H, W, scale, 0.0, batch_size=batch)
        net.propose(p0)
        zs = []
        B = np.array([[0, 0, W - 1.0, H - 1.0]])
        for _ in range(orc.num_levels(H, W) - 1):
            zs.append(net.ctx.head_forward(orc.get_rois_blob(B, scale))[0].ravel())
            B = orc.divide_region(B, 10)
        Tz = float(np.quantile(np.concatenate(zs).astype(np.float64), tzq))
    params = ffi.AzContext.make_params(H, W, scale, Tz, batch_size=batch)
    Y, S, st = net.propose(params, want_scores=True, want_stats=True)
    cfg = orc.OracleCfg(Tz=Tz, BATCH_SIZE=batch)

    class Injected(object):      # pycaffe-shaped view of the HIP head
        name = "inj"
        blobs = net.blobs

        def forward(self, blobs=None, **kw):
            kw.pop("data", None)
            kw["conv5_3"] = fmap
            return net.forward(blobs=blobs, **kw)

    inj = Injected()
    Yref, tr = orc.im_propose({"full": inj, "fc": inj}, (H, W), scale, cfg, return_trace=True)
    # structure: same regions per level, same unique counts, same depth / eval count
    assert st.depth == tr["depth"] and st.num_eval == tr["num_eval"]
    for l, lev in enumerate(tr["levels"]):
        assert st.level_regions[l] == lev["B"].shape[0]
        assert st.level_unique[l] == sum(f["U"] for f in lev["fwd"])
        assert st.level_zoomed[l] == len(lev["indZ"])
    Yall, Sall = net.ctx.last_candidates()
    assert Yall.shape == tr["Y_all"].shape
    assert np.array_equal(Sall.astype(np.float64), tr["aScores"])          # scores: same bits
    np.testing.assert_allclose(Yall, tr["Y_all"], rtol=1e-6, atol=1e-4)    # decode: f32-exp ulps (px)
    # selection: identical candidate indices in identical order (stable ties)
    ref_idx = np.argsort(-tr["aScores"], kind="stable")[:300]
    assert Y.shape == (min(300, Yall.shape[0]), 4)
    assert np.array_equal(Y, Yall[ref_idx])
    assert np.array_equal(S, Sall[ref_idx])
    # NumPy's own (unstable) argsort picks the same boxes up to permutation of exact ties
    np.testing.assert_allclose(np.sort(Yref, axis=0), np.sort(tr["Y_all"][ref_idx], axis=0), rtol=0, atol=0)


def test_fused_loop_vs_cpu_oracle_head(small, mods):
    """End-to-end against the pure-CPU oracle (BLAS head): scores/boxes within 1e-4."""
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    H, W = 600, 1000
    fmap = synth.make_feature_map(5, synth.SMALL_DIMS["C"], 38, 63)
    net.set_conv(fmap)
    params = ffi.AzContext.make_params(H, W, 1.0, 0.0)
    Y, S = net.propose(params, want_scores=True)
    onet = orc.OracleNet(head, feat_fn=lambda d: fmap)
    Yref, tr = orc.im_propose({"full": onet, "fc": onet}, (H, W), 1.0, orc.OracleCfg(Tz=0.0), return_trace=True)
    Yall, Sall = net.ctx.last_candidates()
    assert Yall.shape == tr["Y_all"].shape
    np.testing.assert_allclose(Sall, tr["aScores"], rtol=0, atol=1e-4)
    # px: fp32 head <= 1e-3; the 16-bit-term int6 modes (gemm_mode 2 / 3) keep deltas within 1e-5, which a
    # 1700-px-wide unclipped box turns into <= 2e-2 px
    np.testing.assert_allclose(Yall, tr["Y_all"], rtol=1e-4, atol=2e-2 if net.ctx.gemm_mode else 1e-3)
    # top-300 sets agree except where scores tie within the tolerance
    kth = np.sort(tr["aScores"])[::-1][299]
    sure = tr["aScores"] > kth + 2e-4
    for b in tr["Y_all"][sure]:
        assert np.abs(Y - b).max(axis=1).min() < (3e-2 if net.ctx.gemm_mode else 1e-3)


@pytest.mark.parametrize("H,W,tz", [(600, 1000, 0.0), (375, 500, 0.6), (640, 853, 0.55), (200, 90, 0.0),
                                    (60, 1000, 0.0), (1000, 40, 0.3)])
def test_speculative_levels_are_bit_identical(small, mods, H, W, tz):
    """Levels 1-3: one speculative pass + single-workgroup geometry (default) vs speculative
    pass + separate launches vs level by level: identical bits."""
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    scale = 600.0 / min(H, W)
    if np.round(scale * max(H, W)) > 1000:
        scale = 1000.0 / max(H, W)
    fh, fw = synth.conv_out_size(int(round(H * scale))), synth.conv_out_size(int(round(W * scale)))
    net.set_conv(synth.make_feature_map(6, synth.SMALL_DIMS["C"], fh, fw))
    outs = []
    # (with Tz <= 0 the default is the one-pass plan of az_static.hip: the first variant; the others walk the levels)
    for spec, fused, static in ((True, True, True), (True, True, False), (True, False, False), (False, False, False)):
        p = ffi.AzContext.make_params(H, W, scale, tz, speculate=spec, fused=fused, static_tree=static)
        Y, S, st = net.propose(p, want_scores=True, want_stats=True)
        Ya, Sa = net.ctx.last_candidates()
        outs.append((Y, S, Ya, Sa, list(st.level_regions), list(st.level_unique), list(st.level_zoomed)))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b


def test_topk_kernels_agree(small, mods):
    """Final selection: chip-wide counting kernels vs the single-workgroup radix select."""
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    net.set_conv(synth.make_feature_map(5, synth.SMALL_DIMS["C"], 38, 63))
    for k in (1, 300, 2000, 4096):
        a = net.propose(ffi.AzContext.make_params(600, 1000, 1.0, 0.0, num_proposals=k), want_scores=True)
        b = net.propose(ffi.AzContext.make_params(600, 1000, 1.0, 0.0, num_proposals=k, radix_select=True),
                        want_scores=True)
        assert a[0].shape == (k, 4) and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    rng = np.random.RandomState(2)
    for n in (1, 255, 256, 257, 5000, 40000):
        sc = rng.uniform(0, 1, n).astype(np.float32)
        sc[::5] = sc[0]                                      # ties: lower index first
        got = net.ctx.topk(sc, 300)
        assert np.array_equal(got, np.argsort(-sc, kind="stable")[:300])


def test_graph_replay_equals_direct_launches(small, mods):
    """az_set_graphs: the captured launch sequence gives the same proposals, also when parameters and
    feature maps alternate (one graph per parameter set / map)."""
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    maps = [synth.make_feature_map(s, synth.SMALL_DIMS["C"], 38, 63) for s in (21, 22)]
    plist = [ffi.AzContext.make_params(600, 1000, 1.0, 0.0), ffi.AzContext.make_params(600, 1000, 1.0, 0.0, num_proposals=50),
             ffi.AzContext.make_params(480, 640, 1.25, 0.0, batch_size=100),
             ffi.AzContext.make_params(600, 1000, 1.0, 0.0, fixed_num=False, Tc=0.5)]
    plist[2] = ffi.AzContext.make_params(600, 1000, 1.0, 0.3, batch_size=100)
    want = {}
    for mi, m in enumerate(maps):
        net.set_conv(m)
        for pi, p in enumerate(plist):
            want[(mi, pi)] = net.propose(p, want_scores=True)
    net.ctx.set_graphs(True)
    try:
        for rep in range(3):
            for mi, m in enumerate(maps):
                net.set_conv(m)
                for pi, p in enumerate(plist):
                    got = net.propose(p, want_scores=True)
                    assert np.array_equal(got[0], want[(mi, pi)][0]) and np.array_equal(got[1], want[(mi, pi)][1])
    finally:
        net.ctx.set_graphs(False)


def test_threshold_mode(small, mods):
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    fmap = synth.make_feature_map(5, synth.SMALL_DIMS["C"], 38, 63)
    net.set_conv(fmap)
    params = ffi.AzContext.make_params(600, 1000, 1.0, 0.0, fixed_num=False, Tc=0.6)
    Y, S = net.propose(params, want_scores=True)
    Yall, Sall = net.ctx.last_candidates()
    keep = np.where(Sall.astype(np.float64) >= 0.6)[0]
    assert np.array_equal(Y, Yall[keep]) and np.array_equal(S, Sall[keep])


def test_error_behaviour(small, mods):
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    with pytest.raises(ffi.AzError):
        net.propose(ffi.AzContext.make_params(15, 15, 1.0, 0.0))       # no level fits
    with pytest.raises(ffi.AzError):
        net.ctx.set_feature_map(np.zeros((1, 3, 4, 4), dtype=np.float32))   # wrong channels


def test_full_size_fused_loop(full, mods):
    """BASELINE config: 600x1000, full head, Tz = 0: tree [1,8,32,134,564], 8129 candidates."""
    ffi, synth, HipAZNet, orc = mods
    net, head = full
    fmap = synth.make_feature_map(4, 512, 38, 63)
    net.set_conv(fmap)
    Y, S, st = net.propose(ffi.AzContext.make_params(600, 1000, 1.0, 0.0), want_scores=True, want_stats=True)
    assert list(st.level_regions[:5]) == [1, 8, 32, 134, 564]
    assert list(st.level_unique[:5]) == [1, 8, 32, 130, 517]
    assert st.num_eval == 739 and st.depth == 5 and Y.shape == (300, 4)
    assert np.all(np.diff(S) <= 0)                      # sorted by score
    assert np.all(Y[:, 0] >= 0) and np.all(Y[:, 2] <= 999) and np.all(Y[:, 3] <= 599)
    assert np.all(np.minimum(Y[:, 2] - Y[:, 0], Y[:, 3] - Y[:, 1]) + 1 >= 10)
    # per-level head vs the fused loop on the level-4 regions: same bits
    B = np.array([[0, 0, 999.0, 599.0]])
    for _ in range(3):
        B = orc.divide_region(B, 10)
    rois = orc.get_rois_blob(B, 1.0)
    idx, inv = orc.roi_dedup(rois)
    z, p, d = net.ctx.head_forward(rois[idx])
    zr, pr, dr = orc.head_forward(head, fmap[0], rois[idx][:16])
    np.testing.assert_allclose(z[:16], zr, rtol=0, atol=1e-4)
    np.testing.assert_allclose(p[:16], pr, rtol=0, atol=1e-4)


# ---------------------------------------------------------------- Fast R-CNN head (config 3)
@pytest.fixture(scope="module")
def det_small(small, mods):
    ffi, synth, HipAZNet, orc = mods
    from aznet_hip.net import HipDetNet
    net, head = small
    dhead = synth.make_det_head(seed=99, **synth.SMALL_DET_DIMS)
    return HipDetNet(dhead, net), dhead


@pytest.mark.parametrize("R", [1, 40, 300])
def test_det_head_forward(det_small, small, mods, R):
    ffi, synth, HipAZNet, orc = mods
    dnet, dhead = det_small
    net, head = small
    fmap = synth.make_feature_map(8, synth.SMALL_DIMS["C"], 38, 63)
    net.set_conv(fmap)
    rois = _rand_rois(np.random.RandomState(R + 5), R, 1000, 600)
    p, b = net.ctx.det_forward(rois)
    pr, br = orc.det_head_forward(dhead, fmap[0], rois)
    np.testing.assert_allclose(p, pr, rtol=0, atol=1e-4)
    np.testing.assert_allclose(b, br, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(p.sum(1), 1.0, rtol=0, atol=1e-5)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_detect_vs_oracle_loop(det_small, small, mods, tag):
    """az_detect (dedup + head + per-class decode + un-dedup in one call) against the oracle's
    _frcnn_forward driven by the HIP detection head through the pycaffe-shaped seam, on the
    proposals of the reference's own run."""
    ffi, synth, HipAZNet, orc = mods
    dnet, dhead = det_small
    net, head = small
    g = load("g9_detect_%s.npz" % tag)
    H, W, scale, batch = int(g["H"]), int(g["W"]), float(g["scale"]), int(g["batch"])
    fmap = synth.make_feature_map(8, synth.SMALL_DIMS["C"], synth.conv_out_size(int(round(H * scale))),
                                  synth.conv_out_size(int(round(W * scale))))
    net.set_conv(fmap)
    props = g["proposals"]
    s, b = net.ctx.detect(props, scale, H, W, batch_size=batch)
    cfg = orc.OracleCfg(BATCH_SIZE=batch)

    class Inj(object):
        blobs = dnet.blobs

        def forward(self, blobs=None, **kw):
            kw["conv5_3"] = fmap
            return dnet.forward(blobs=blobs, **kw)

    sr, br = orc.frcnn_forward({"fc": Inj()}, (H, W), scale, props, 21, {"conv5_3": fmap}, cfg)
    assert s.shape == (props.shape[0], 21) and b.shape == (props.shape[0], 84)
    assert np.array_equal(s.astype(np.float64), sr)                 # same head, same bits
    np.testing.assert_allclose(b, br, rtol=1e-6, atol=1e-4)           # decode: f32-exp ulps, px
    # and against the recorded CPU run of the reference (BLAS head): within tolerance
    np.testing.assert_allclose(s, g["scores"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(b, g["pred_boxes"], rtol=1e-4, atol=1e-3)


def test_im_detect_shared_and_apply_nms(det_small, small, mods):
    """Host API of config 3: im_detect_shared -> per-class boxes, apply_nms through az_nms."""
    ffi, synth, HipAZNet, orc = mods
    from detect import test as T
    from detect.config import cfg, cfg_set_mode
    dnet, dhead = det_small
    net, head = small
    cfg_set_mode("Test", 0.0)
    fmap = synth.make_feature_map(8, synth.SMALL_DIMS["C"], 38, 63)
    im = synth.make_image(4, 600, 1000)
    boxes = T.im_propose(net, im, conv={"conv5_3": fmap})
    scores, pred = T.im_detect(dnet, im, boxes, 21)
    assert scores.shape == (300, 21) and pred.shape == (300, 84) and scores.dtype == np.float64
    all_boxes = [[[]] for _ in range(21)]
    for j in range(1, 21):
        top = np.argsort(-scores[:, j])[:100]
        all_boxes[j][0] = np.hstack((pred[top, 4 * j:4 * j + 4], scores[top, j:j + 1])).astype(np.float32)
    nmsd = T.apply_nms(all_boxes, 0.5)
    for j in range(1, 21):
        ref = orc.nms(all_boxes[j][0], 0.5)
        assert np.array_equal(nmsd[j][0], all_boxes[j][0][ref])


# ---------------------------------------------------------------- int6 on the 16-bit matrix cores (gemm_mode 2 / 3)
@pytest.mark.parametrize("mode", [2, 3])
def test_16bit_term_modes_accuracy_and_row_independence(mods, mode):
    """az_set_gemm_mode(2 / 3): int6 on the fp16 / bf16 matrix cores with fp32 operands as two fp16 resp. three bf16
    terms (reduced head here; the full head, against f64: tests/test_gpu_gemm_modes.py).  Outputs within 1e-4 of the
    fp32 path and of the BLAS oracle (measured ~1e-6), and a roi's bits do not depend on the batch (both tile
    shapes use the same per-row arithmetic)."""
    ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    fmap = synth.make_feature_map(3, synth.SMALL_DIMS["C"], 38, 63)
    rois = _rand_rois(np.random.RandomState(11), 300, 1000, 600)
    ref = orc.head_forward(head, fmap[0], rois)
    nets = {m: HipAZNet(head, name="m%d" % m, max_regions=1024, gemm_mode=m) for m in (0, mode)}
    out = {}
    for m, net in nets.items():
        net.set_conv(fmap)
        out[m] = net.ctx.head_forward(rois)
    for a, b, r in zip(out[mode], out[0], ref):
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(a, r, rtol=1e-4, atol=1e-4)
    for n in (1, 40, 64, 65, 130):          # 1-2 strips: 4-wave shape; >= 3 strips: 8-wave shape
        sub = nets[mode].ctx.head_forward(rois[:n])
        for a, b in zip(sub, out[mode]):
            assert np.array_equal(a, b[:n])
    # whole search vs mode 0: same tree, scores within 1e-4
    p = ffi.AzContext.make_params(600, 1000, 1.0, 0.0)
    res = {m: nets[m].propose(p, want_scores=True, want_stats=True) for m in (0, mode)}
    assert list(res[0][2].level_unique[:5]) == list(res[mode][2].level_unique[:5])
    c0, c2 = nets[0].ctx.last_candidates(), nets[mode].ctx.last_candidates()
    assert c0[0].shape == c2[0].shape
    np.testing.assert_allclose(c2[1], c0[1], rtol=0, atol=1e-4)
