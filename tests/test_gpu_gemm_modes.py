"""int6 on the 16-bit matrix cores (az_set_gemm_mode, include/aznet_hip.h): fp32 operands as two fp16 terms (mode 2,
3 MFMAs per product) or three bf16 terms (mode 3, 6 MFMAs), fp32 accumulation.

What is checked, at the full head (25088 -> 4096 -> ...):
  * the head's outputs against an F64 evaluation of the same head (numpy, on the oracle's RoIPool): the error of
    modes 2 and 3 is of the size of mode 0's (the fp32-MFMA path) and of the fp32 BLAS oracle's -- they give up
    nothing the reference's fp32 arithmetic has;
  * a roi's bits do not depend on the batch it is evaluated in (1 ... 300 rows: both kernel shapes);
  * maps whose values are tiny, huge or signed (mode 2 scales the fp16 terms per image by a power of two);
  * whole searches (level loop, one pass, calibrated Tz) against the pure-CPU oracle, tree exact, scores 1e-4;
  * two searches queued on one context with maps of different magnitude (the scale belongs to the image)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from aznet_hip import ffi, synth
    from aznet_hip.net import HipAZNet
    from oracle import az_oracle as orc
    return ffi, synth, HipAZNet, orc


@pytest.fixture(scope="module")
def full(mods):
    ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=1234, **synth.FULL_DIMS)
    nets = {m: HipAZNet(head, name="mode%d" % m, max_regions=4096, gemm_mode=m) for m in (0, 2, 3)}
    return head, nets


def _rois(n, seed=3):
    rng = np.random.RandomState(seed)
    x1 = rng.uniform(0, 900, n); y1 = rng.uniform(0, 500, n)
    return np.stack([np.zeros(n), x1, y1, x1 + rng.uniform(16, 300, n), y1 + rng.uniform(16, 300, n)], 1).astype(np.float32)


def _f64_head(orc, head, fmap, rois):
    n = rois.shape[0]
    p5 = orc.roi_pool(fmap[0], rois).reshape(n, -1).astype(np.float64)

    def f(x, W, b, relu):
        y = x @ W.astype(np.float64).T + b.astype(np.float64)
        return np.maximum(y, 0) if relu else y
    h6 = f(p5, head["W6"], head["b6"], True)
    h71 = f(h6, head["W71"], head["b71"], True)
    h72 = f(h6, head["W72"], head["b72"], True)
    sg = lambda x: 1.0 / (1.0 + np.exp(-x))          # noqa: E731
    return sg(f(h72, head["Wz"], head["bz"], False)), sg(f(h71, head["Was"], head["bas"], False)), \
        f(h71, head["Wab"], head["bab"], False)


@pytest.mark.parametrize("gain", [1.0, 1e-4, 3e4, -1.0], ids=["as_is", "tiny", "huge", "negated"])
def test_outputs_are_as_close_to_f64_as_the_fp32_path(full, mods, gain):
    ffi, synth, HipAZNet, orc = mods
    head, nets = full
    fmap = (synth.make_feature_map(4, 512, 38, 63) * np.float32(gain)).astype(np.float32)
    rois = _rois(96)
    truth = _f64_head(orc, head, fmap, rois)
    err = {}
    for m, net in nets.items():
        net.set_conv(fmap)
        out = net.ctx.head_forward(rois)
        # (relative to the size of the outputs: the deltas of the huge map are huge)
        err[m] = [np.abs(a - t).max() / max(1.0, np.abs(t).max()) for a, t in zip(out, truth)]
    for m in (2, 3):
        for e, e0 in zip(err[m], err[0]):
            if abs(gain) <= 1.0:
                assert e <= 1e-5, (m, err)         # an order inside the 1e-4 tolerance
            # (the huge map drives the pre-sigmoid values to +-1e5: every path's score error grows with them)
            assert e <= 3.0 * e0 + 2e-7, (m, err)  # of the size of the fp32-MFMA path's own error


def test_a_rois_bits_do_not_depend_on_its_batch(full, mods):
    head, nets = full
    ffi, synth, HipAZNet, orc = mods
    fmap = synth.make_feature_map(5, 512, 38, 63)
    rois = _rois(300, seed=8)
    for m in (2, 3):
        nets[m].set_conv(fmap)
        ref = nets[m].ctx.head_forward(rois)
        for n in (1, 40, 64, 65, 130, 257):          # <= 2 strips: the 4-wave shape; more: the 8-wave shape, 1-2 m-tiles
            sub = nets[m].ctx.head_forward(rois[:n])
            for a, b in zip(sub, ref):
                assert np.array_equal(a, b[:n]), (m, n)


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("form", ["level_loop", "one_pass", "calibrated"])
def test_search_vs_pure_cpu_oracle(full, mods, mode, form):
    ffi, synth, HipAZNet, orc = mods
    head, nets = full
    net = nets[mode]
    H, W = 600, 1000
    fmap = synth.make_feature_map(31, 512, 38, 63)
    net.set_conv(fmap)
    onet = orc.OracleNet(head, feat_fn=lambda d: fmap)
    onets = {"full": onet, "fc": onet}
    Tz = 0.0
    if form == "calibrated":
        _, tr0 = orc.im_propose(onets, (H, W), 1.0, orc.OracleCfg(Tz=0.0), return_trace=True)
        zs = np.sort(np.concatenate([lv["zoom"] for lv in tr0["levels"][1:3]]))
        k = len(zs) // 2
        j = next(j for j in range(k, len(zs) - 1) if zs[j + 1] - zs[j] > 2e-3)
        Tz = 0.5 * (zs[j] + zs[j + 1])
    Yref, tr = orc.im_propose(onets, (H, W), 1.0, orc.OracleCfg(Tz=Tz), return_trace=True)
    z = np.concatenate([lv["zoom"] for lv in tr["levels"]])
    assert np.abs(z - Tz).min() > 2e-4
    p = ffi.AzContext.make_params(H, W, 1.0, Tz, static_tree=(form == "one_pass"))
    Y, S, st = net.propose(p, want_scores=True, want_stats=True)
    assert st.static_plan == (1 if form == "one_pass" else 0)
    assert st.depth == tr["depth"] and st.num_eval == tr["num_eval"]
    for l, lev in enumerate(tr["levels"]):
        assert st.level_regions[l] == lev["B"].shape[0]
        assert st.level_unique[l] == sum(f["U"] for f in lev["fwd"])
        assert st.level_zoomed[l] == len(lev["indZ"])
    Yall, Sall = net.ctx.last_candidates()
    assert Yall.shape == tr["Y_all"].shape
    assert np.abs(Sall.astype(np.float64) - tr["aScores"]).max() <= 1e-4
    np.testing.assert_allclose(Yall, tr["Y_all"], rtol=1e-4, atol=2e-2)
    k = min(300, tr["Y_all"].shape[0])
    assert Y.shape == (k, 4)
    sure = tr["aScores"] > np.sort(tr["aScores"])[::-1][k - 1] + 2e-4 if k < tr["Y_all"].shape[0] else np.ones(k, bool)
    for b in tr["Y_all"][sure]:
        assert np.abs(Y - b).max(axis=1).min() <= 2e-2


def test_queued_searches_keep_their_own_scale(full, mods):
    """mode 2: the power-of-two scale of the fp16 terms is a property of the image; two searches in flight on one
    context, maps of very different magnitude, each equal to the same search run alone."""
    ffi, synth, HipAZNet, orc = mods
    head, nets = full
    net = nets[2]
    maps = [synth.make_feature_map(41, 512, 38, 63), (synth.make_feature_map(42, 512, 38, 63) * np.float32(37.0))]
    p = ffi.AzContext.make_params(600, 1000, 1.0, 0.0, static_tree=False)
    alone = []
    for m in maps:
        net.set_conv(m)
        alone.append(net.propose(p, want_scores=True))
    import torch
    dev = [torch.from_numpy(m).cuda() for m in maps]
    for rnd in range(3):
        net.ctx.propose_launch(p, fmap=dev[0], producer_done=True)
        net.ctx.propose_launch(p, fmap=dev[1], producer_done=True)
        for i in range(2):
            Y, S = net.ctx.propose_fetch(want_scores=True)
            assert np.array_equal(Y, alone[i][0]) and np.array_equal(S, alone[i][1]), (rnd, i)


def test_mode_values(mods):
    ffi = mods[0]
    for bad in (1, 4, 5):
        with pytest.raises(ValueError):
            ffi.AzContext(0, gemm_mode=bad)


# ---- adversarial operands for the exact three-bf16-term split (mode 3) against the fp32-MFMA path (mode 0) ---------------
# A head built so that int6's PRE-activation sums come out of az_head_forward untouched: W6 holds 44 weight rows w_j and
# their negatives, int7_1 is the identity on those 88 units and adj_bbox_j = relu(y_j) - relu(-y_j) = y_j (one of the two is
# 0; every other weight is 0.0, and x + 0 is exact), so adj_bbox IS x . W6^T as the GEMM computed it.
def _probe_head(w_rows):
    n6, n71, n72 = 128, 88, 4
    K6 = w_rows.shape[1]
    W6 = np.zeros((n6, K6), np.float32)
    W6[0:88:2] = w_rows
    W6[1:88:2] = -w_rows
    W71 = np.zeros((n71, n6), np.float32)
    W71[np.arange(88), np.arange(88)] = 1.0
    Wab = np.zeros((44, n71), np.float32)
    Wab[np.arange(44), 2 * np.arange(44)] = 1.0
    Wab[np.arange(44), 2 * np.arange(44) + 1] = -1.0
    z = lambda *s: np.zeros(s, np.float32)          # noqa: E731
    return {"W6": W6, "b6": z(n6), "W71": W71, "b71": z(n71), "W72": z(n72, n6), "b72": z(n72), "Was": z(11, n71),
            "bas": z(11), "Wab": Wab, "bab": z(44), "Wz": z(1, n72), "bz": z(1)}


def _probe_rois(n, seed, hw=256):
    rng = np.random.RandomState(seed)
    x1 = rng.uniform(0, hw - 80, n); y1 = rng.uniform(0, hw - 80, n)
    return np.stack([np.zeros(n), x1, y1, x1 + rng.uniform(16, 70, n), y1 + rng.uniform(16, 70, n)], 1).astype(np.float32)


def _int6_sums(HipAZNet, head, fmap, rois, mode):
    net = HipAZNet(head, name="probe%d" % mode, gemm_mode=mode)
    net.set_conv(fmap)
    p5 = net.ctx.roi_pool(rois)                                     # (bit-exact kernel: the GEMM's A operand)
    return net.ctx.head_forward(rois)[2].astype(np.float64), p5.astype(np.float64)


def test_exact_split_on_adversarial_operands_is_as_good_as_fp32_mfma(mods):
    """|x| and |w| spanning 2^+-60 (products of order one), subnormal activations, rows built to cancel: the error of the
    three-bf16-term path against an f64 evaluation, relative to sum |w||x| (the forward-error scale of a dot product), is
    of the size of the fp32-MFMA path's on the same operands -- the split gives up neither mantissa bits nor exponent
    range.  (Mode 2's fp16 terms have no such range: it is not in this test.)"""
    ffi, synth, HipAZNet, orc = mods
    C, HW = 16, 16
    rng = np.random.RandomState(11)
    fmap = rng.standard_normal((1, C, HW, HW)).astype(np.float32)
    w = rng.standard_normal((44, C, 49)).astype(np.float32)
    big, tiny = np.float32(2.0 ** 60), np.float32(2.0 ** -60)
    fmap[0, 0:4] *= big; w[:, 0:4] *= tiny                          # huge activations, tiny weights
    fmap[0, 4:8] *= tiny; w[:, 4:8] *= big                          # and the other way round
    fmap[0, 12] = np.abs(fmap[0, 12]) * np.float32(1e6)             # channels 12 / 13: equal activations, opposite weights --
    fmap[0, 13] = fmap[0, 12]                                       # 1e6-sized terms that cancel to nothing
    w[:, 13] = -w[:, 12]
    fmap[0, 14:16] = (np.abs(fmap[0, 14:16]) * np.float32(1e-39)).astype(np.float32)     # subnormal activations
    assert 0 < fmap[0, 14].max() < np.finfo(np.float32).tiny
    head = _probe_head(w.reshape(44, C * 49))
    rois = _probe_rois(40, 5)
    err = {}
    for mode in (0, 3):
        y, p5 = _int6_sums(HipAZNet, head, fmap, rois, mode)
        W = head["W6"][0:88:2].astype(np.float64)
        truth = p5 @ W.T
        scale = np.abs(p5) @ np.abs(W).T
        assert np.isfinite(y).all() and (scale > 0).all()
        err[mode] = float((np.abs(y - truth) / scale).max())
    # (fp32 accumulation of 784 terms: a few 1e-7 of the scale either way)
    assert err[0] < 2e-6 and err[3] < 2e-6, err
    assert err[3] <= 2.0 * err[0] + 2.0 ** -22, err


def test_a_non_finite_activation_stays_in_the_rois_that_see_it(mods):
    """An inf or NaN in the map (a conv5_3 map is finite; this is the guard rail): on both paths every roi whose windows do
    not hold the cell is bit-identical to the same map without it.  What the rois that DO see it get differs and is not part of
    the contract: fp32 arithmetic carries +-inf through int6, the terms of a split inf are (inf, inf - inf = NaN), and the
    ReLU behind int6 is an fmax, which returns the other operand for a NaN -- documented in include/aznet_hip.h."""
    ffi, synth, HipAZNet, orc = mods
    C, HW = 16, 16
    rng = np.random.RandomState(12)
    base = np.abs(rng.standard_normal((1, C, HW, HW))).astype(np.float32)
    head = _probe_head(rng.standard_normal((44, C * 49)).astype(np.float32))
    rois = _probe_rois(40, 6)
    for bad in (np.inf, np.nan):
        fmap = base.copy()
        fmap[0, 9, 5, 7] = bad                                       # one cell of one channel
        for mode in (0, 3):
            y, p5 = _int6_sums(HipAZNet, head, fmap, rois, mode)
            y0, _ = _int6_sums(HipAZNet, head, base, rois, mode)
            hit = ~np.isfinite(p5).all(axis=1)
            if bad is np.nan:
                # (RoIPool's max is `if (x > best)`, as Caffe's: a NaN cell never wins a bin -- nothing non-finite gets out)
                assert hit.sum() == 0 and np.isfinite(y).all()
                continue
            assert 0 < hit.sum() < len(rois)
            assert np.isfinite(y0).all()
            assert np.array_equal(y[~hit], y0[~hit]), (bad, mode)
