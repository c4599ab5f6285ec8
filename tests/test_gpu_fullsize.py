"""GPU parity at BASELINE's full sizes, against the PURE-CPU oracle (no HIP head in the checker):

  * the full-size head (25088 -> 4096 -> {1024 -> 11+44, 256 -> 1}) at the row counts the benchmark
    launches -- 517 (level 5), 130 (level 4), the 49-row speculative pass -- and around them;
  * config A end to end (600x1000, full head, Tz = 0 and a calibrated Tz): candidates, scores and the
    top-300 against oracle.im_propose driven by the BLAS head;
  * the reference's own recorded runs (tests/golden/g7_trace_*.npz: lib/detect/test.py:im_propose with the
    seed-77 head on the seed-5 map) fed to az_propose: per-level rois bit-exact, final Y within tolerance;
  * cfg.DEDUP_BOXES <= 0 (no dedup);
  * the Fast R-CNN head at its real size (4096/4096/21, BASELINE config 3).

Tolerances (north_star): integer / geometry work bit-exact; scores and box deltas 1e-4; pixel boxes
1e-4 relative to the box scale (a delta error of 1e-4 on a 1000-px anchor is 0.1 px: the bound used
is 2e-2 px, measured values are ~1e-3)."""
import numpy as np
import pytest

from helpers import load, replay_nets, TRACES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from aznet_hip import ffi, synth
    from aznet_hip.net import HipAZNet
    from oracle import az_oracle as orc
    return ffi, synth, HipAZNet, orc


@pytest.fixture(scope="module")
def full(mods, gemm_mode):
    ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=1234, **synth.FULL_DIMS)
    net = HipAZNet(head, name="full", max_regions=4096, gemm_mode=gemm_mode)
    fmap = synth.make_feature_map(4, 512, 38, 63)
    net.set_conv(fmap)
    return net, head, fmap


@pytest.fixture(scope="module")
def small(mods):
    ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    return HipAZNet(head, name="small"), head


def _level_rois(orc, H, W, scale, depth):
    """Unique rois of level `depth` (1-based) of the full tree, in np.unique order (what the GPU forwards)."""
    B = np.array([[0, 0, W - 1.0, H - 1.0]])
    for _ in range(depth - 1):
        B = orc.divide_region(B, 10)
    rois = orc.get_rois_blob(B, scale)
    idx, inv = orc.roi_dedup(rois)
    return rois[idx]


@pytest.mark.parametrize("R", [517, 564, 49, 41, 130, 16, 17, 48])
def test_full_head_vs_cpu_oracle_at_launch_sizes(full, mods, R):
    """The dominant launch shapes (U = 517: five m-tiles + a 5-row half strip, K = 25088; 130; the 49-row
    speculative pass) and their neighbours: zoom / adjacency scores and box deltas within 1e-4 of the
    CPU (BLAS) oracle on the SAME rois the search forwards."""
    ffi, synth, HipAZNet, orc = mods
    net, head, fmap = full
    net.set_conv(fmap)
    if R == 517:
        rois = _level_rois(orc, 600, 1000, 1.0, 5)
    elif R == 130:
        rois = _level_rois(orc, 600, 1000, 1.0, 4)
    else:
        rng = np.random.RandomState(R)
        lv = np.vstack([_level_rois(orc, 600, 1000, 1.0, d) for d in (1, 2, 3, 4, 5)])
        rois = lv[rng.permutation(lv.shape[0])[:R]] if R <= lv.shape[0] else lv
        if R > lv.shape[0]:
            rois = np.vstack([lv, lv[:R - lv.shape[0]] + np.float32([0, 3, 5, -4, -2])])
    assert rois.shape == (R, 5)
    z, p, d = net.ctx.head_forward(rois)
    zr, pr, dr = orc.head_forward(head, fmap[0], rois)
    assert np.abs(z - zr).max() <= 1e-4 and np.abs(p - pr).max() <= 1e-4
    np.testing.assert_allclose(d, dr, rtol=1e-4, atol=1e-4)


def _calibrated_tz(z_all, q):
    """A threshold at quantile q of the CPU zoom scores, moved clear of every score by >= 1e-3 so that
    fp32 summation-order differences between the GPU and the BLAS oracle cannot flip a zoom decision."""
    zs = np.sort(z_all.astype(np.float64))
    k = int(q * (len(zs) - 1))
    for j in range(k, len(zs) - 1):
        if zs[j + 1] - zs[j] > 2e-3:
            return 0.5 * (zs[j] + zs[j + 1])
    raise AssertionError("no gap in the zoom scores")


@pytest.mark.parametrize("mode", ["tz0", "tz0_level_loop", "calibrated"])
def test_config_a_full_head_vs_pure_cpu_oracle(full, mods, mode):
    """BASELINE config A (600x1000, full head) through az_propose vs the oracle's whole loop on the CPU.
    tz0: all levels in one head pass (the default when Tz <= 0); tz0_level_loop: the same search level by level."""
    ffi, synth, HipAZNet, orc = mods
    net, head, fmap = full
    net.set_conv(fmap)
    H, W = 600, 1000
    onet = orc.OracleNet(head, feat_fn=lambda d: fmap)
    nets = {"full": onet, "fc": onet}
    Tz = 0.0
    if mode == "calibrated":
        _, tr0 = orc.im_propose(nets, (H, W), 1.0, orc.OracleCfg(Tz=0.0), return_trace=True)
        Tz = _calibrated_tz(np.concatenate([lv["zoom"][1:] if i == 0 else lv["zoom"]
                                            for i, lv in enumerate(tr0["levels"][:3])]), 0.5)
    Yref, tr = orc.im_propose(nets, (H, W), 1.0, orc.OracleCfg(Tz=Tz), return_trace=True)
    Y, S, st = net.propose(ffi.AzContext.make_params(H, W, 1.0, Tz, static_tree=(mode != "tz0_level_loop")),
                           want_scores=True, want_stats=True)
    assert st.static_plan == (1 if mode == "tz0" else 0)
    # (48: the root's row deferred to level 4's pass -- taken when the context's last searches all reached that level; 49 otherwise)
    assert st.spec_rows == 688 if mode == "tz0" else st.spec_rows in (48, 49)
    # tree: same regions per level, same unique counts, same zoom sets (integer work: exact)
    assert st.depth == tr["depth"] and st.num_eval == tr["num_eval"]
    for l, lev in enumerate(tr["levels"]):
        assert st.level_regions[l] == lev["B"].shape[0]
        assert st.level_unique[l] == sum(f["U"] for f in lev["fwd"])
        assert st.level_zoomed[l] == len(lev["indZ"])
    if mode != "calibrated":
        assert [int(st.level_regions[l]) for l in range(5)] == [1, 8, 32, 134, 564]
        assert [int(st.level_unique[l]) for l in range(5)] == [1, 8, 32, 130, 517]
    else:
        assert st.num_eval < 739                               # a partially expanded tree
    Yall, Sall = net.ctx.last_candidates()
    assert Yall.shape == tr["Y_all"].shape
    assert np.abs(Sall.astype(np.float64) - tr["aScores"]).max() <= 1e-4
    np.testing.assert_allclose(Yall, tr["Y_all"], rtol=1e-4, atol=2e-2)
    # top-300: every reference proposal whose score clears the 300th by more than the tolerance is returned
    assert Y.shape == Yref.shape == (300, 4)
    order = np.sort(tr["aScores"])[::-1]
    sure = tr["aScores"] > order[299] + 2e-4
    assert sure.sum() >= 250
    for b in tr["Y_all"][sure]:
        assert np.abs(Y - b).max(axis=1).min() <= 2e-2
    assert np.all(np.diff(S) <= 0)


@pytest.mark.parametrize("tag", TRACES)
def test_reference_traces_through_az_propose(small, mods, tag):
    """g7: what the REFERENCE's im_propose did (recorded by oracle/gen_golden.py from lib/detect/test.py
    itself, head = seed-77 small head on the CPU, map = seed-5): az_propose on the same inputs must forward
    the same rois at every level (bit-exact f32) and return the reference's Y."""
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    g = load("g7_trace_%s.npz" % tag)
    H, W, Tz, batch, scale = int(g["H"]), int(g["W"]), float(g["Tz"]), int(g["batch"]), float(g["scale"])
    shp = [int(x) for x in g["fmap_shape"]]
    fmap = synth.make_feature_map(5, shp[1], shp[2], shp[3])
    net.set_conv(fmap)
    Y, S, st = net.propose(ffi.AzContext.make_params(H, W, scale, Tz, batch_size=batch), want_scores=True,
                           want_stats=True)
    Yall, Sall = net.ctx.last_candidates()          # (before the unit calls below reuse the buffers)
    # the oracle loop replaying the reference's recorded head outputs reproduces the reference's Y
    # (tests/test_oracle_golden.py); its trace gives the per-level structure of the reference run
    Yrep, tr = orc.im_propose(replay_nets(g), (H, W), scale, orc.OracleCfg(Tz=Tz, BATCH_SIZE=batch),
                              return_trace=True)
    np.testing.assert_allclose(Yrep, g["Y"], rtol=1e-6, atol=1e-9)      # (np.exp f32 may differ by an ulp across hosts)
    assert st.depth == tr["depth"] and st.num_eval == tr["num_eval"]
    for l, lev in enumerate(tr["levels"]):
        assert st.level_regions[l] == lev["B"].shape[0]
        assert st.level_unique[l] == sum(f["U"] for f in lev["fwd"])
        assert st.level_zoomed[l] == len(lev["indZ"])
    # rois the reference fed to Caffe at each call == the GPU's projection + dedup of the same level
    ncalls = int(g["ncalls"])
    calls = sorted(range(ncalls), key=lambda i: (not bool(g["c%d_full" % i]), i))     # 'full' call first
    ci = 0
    for lev in tr["levels"]:
        rois, index, inv = net.ctx.roi_dedup(lev["B"], scale, 1. / 16., batch)
        got = rois[index]
        ref = np.vstack([g["c%d_rois" % calls[ci + j]] for j in range(len(lev["fwd"]))])
        ci += len(lev["fwd"])
        assert got.dtype == np.float32 and np.array_equal(got, ref)
    assert ci == ncalls
    # candidates and final proposals vs the reference's run (its head ran on the CPU: 1e-4)
    assert Yall.shape == tr["Y_all"].shape
    assert np.abs(Sall.astype(np.float64) - tr["aScores"]).max() <= 1e-4
    np.testing.assert_allclose(Yall, tr["Y_all"], rtol=1e-4, atol=1e-3)
    assert Y.shape == g["Y"].shape
    k = Y.shape[0]
    if k == tr["Y_all"].shape[0]:
        sure = np.ones(k, dtype=bool)
    else:
        sure = tr["aScores"] > np.sort(tr["aScores"])[::-1][k - 1] + 2e-4
    for b in tr["Y_all"][sure]:
        assert np.abs(Y - b).max(axis=1).min() <= 1e-3


@pytest.mark.parametrize("H,W,batch", [(600, 1000, 10000), (375, 500, 50)])
def test_no_dedup_when_dedup_boxes_is_zero(small, mods, H, W, batch):
    """cfg.DEDUP_BOXES <= 0: `if cfg.DEDUP_BOXES > 0:` (lib/detect/test.py:211,246) skips the feature-space
    dedup -- every region is forwarded, index = inv_index = identity."""
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    scale = 600.0 / min(H, W)
    fh, fw = synth.conv_out_size(int(round(H * scale))), synth.conv_out_size(int(round(W * scale)))
    fmap = synth.make_feature_map(7, synth.SMALL_DIMS["C"], fh, fw)
    net.set_conv(fmap)
    B = np.array([[0, 0, W - 1.0, H - 1.0]])
    for _ in range(3):
        B = orc.divide_region(B, 10)
    rois, index, inv = net.ctx.roi_dedup(B, scale, 0.0, batch)
    assert np.array_equal(index, np.arange(B.shape[0])) and np.array_equal(inv, np.arange(B.shape[0]))
    assert np.array_equal(rois, orc.get_rois_blob(B, scale))
    outs = []
    for spec, fused in ((True, True), (False, False)):
        p = ffi.AzContext.make_params(H, W, scale, 0.0, dedup=0.0, batch_size=batch, speculate=spec, fused=fused)
        Y, S, st = net.propose(p, want_scores=True, want_stats=True)
        assert [int(st.level_unique[l]) for l in range(st.n_levels)] == [int(st.level_regions[l]) for l in range(st.n_levels)]
        outs.append((Y, S) + net.ctx.last_candidates())
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    onet = orc.OracleNet(head, feat_fn=lambda d: fmap)
    Yref, tr = orc.im_propose({"full": onet, "fc": onet}, (H, W), scale,
                              orc.OracleCfg(Tz=0.0, DEDUP_BOXES=0.0, BATCH_SIZE=batch), return_trace=True)
    assert [sum(f["U"] for f in lv["fwd"]) for lv in tr["levels"]] == [lv["B"].shape[0] for lv in tr["levels"]]
    Yall, Sall = outs[0][2], outs[0][3]
    assert Yall.shape == tr["Y_all"].shape
    assert np.abs(Sall.astype(np.float64) - tr["aScores"]).max() <= 1e-4
    np.testing.assert_allclose(Yall, tr["Y_all"], rtol=1e-4, atol=1e-3)
    # with the dedup on, the same tree is searched (the tree does not depend on the head at Tz = 0)
    net.propose(ffi.AzContext.make_params(H, W, scale, 0.0, batch_size=batch))
    assert net.ctx.last_candidates()[0].shape[0] > 0


def test_last_candidates_survives_counter_reuse_but_not_buffer_reuse(small, mods):
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    net.set_conv(synth.make_feature_map(7, synth.SMALL_DIMS["C"], 38, 63))
    net.propose(ffi.AzContext.make_params(600, 1000, 1.0, 0.0))
    a = net.ctx.last_candidates()
    assert a[0].shape[0] > 300
    net.ctx.divide_region(np.array([[0, 0, 999.0, 599.0]]), 10.0)        # reuses the counters, not the candidates
    b = net.ctx.last_candidates()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    net.ctx.topk(np.arange(10, dtype=np.float32), 3)                     # overwrites the candidate scores
    with pytest.raises(ffi.AzError):
        net.ctx.last_candidates()


# ---------------------------------------------------------------- config 3 at its real size
@pytest.mark.parametrize("ncls", [21, 81], ids=["voc21", "coco81"])
def test_full_size_det_head_vs_cpu_oracle(full, mods, ncls):
    """Fast R-CNN head at the size of models/Pascal/VGG16/frcnn/test_fc.prototxt (fc6/fc7 4096, 21 classes) and of
    models/COCO/VGG16/frcnn/test_fc.prototxt:97-135 (81 classes: cls_score 81, bbox_pred 324)
    on the 300 proposals of a config-A search, vs the CPU oracle: cls_prob 1e-4, bbox_pred deltas 1e-4;
    az_detect (dedup + decode + un-dedup) vs oracle.frcnn_forward."""
    ffi, synth, HipAZNet, orc = mods
    from aznet_hip.net import HipDetNet
    net, head, fmap = full
    net.set_conv(fmap)
    dhead = synth.make_det_head(seed=4242, **dict(synth.FULL_DET_DIMS, ncls=ncls))
    dnet = HipDetNet(dhead, net)
    H, W = 600, 1000
    props = net.propose(ffi.AzContext.make_params(H, W, 1.0, 0.0))
    assert props.shape == (300, 4)
    rois = orc.get_rois_blob(props, 1.0)
    idx, inv = orc.roi_dedup(rois)
    p, b = net.ctx.det_forward(rois[idx])
    pr, br = orc.det_head_forward(dhead, fmap[0], rois[idx])
    assert np.abs(p - pr).max() <= 1e-4
    np.testing.assert_allclose(b, br, rtol=1e-4, atol=1e-4)
    s, bx = net.ctx.detect(props, 1.0, H, W)
    odet = orc.OracleDetNet(dhead)
    sr, bxr = orc.frcnn_forward({"fc": odet}, (H, W), 1.0, props, ncls, {"conv5_3": fmap}, orc.OracleCfg())
    assert s.shape == (300, ncls) and bx.shape == (300, 4 * ncls)
    assert np.abs(s.sum(axis=1) - 1.0).max() <= 1e-5
    assert np.abs(s - sr).max() <= 1e-4
    np.testing.assert_allclose(bx, bxr, rtol=1e-4, atol=2e-2)


@pytest.mark.parametrize("ncls", [2, 63, 64, 65, 128, 129, 256])
def test_det_head_class_counts_across_the_lane_boundaries(small, mods, ncls):
    """k_det_epilogue gives a lane the classes l, l + 64, ...: class counts at and around the multiples of 64, softmax sum in
    channel order, every class box decoded -- vs the CPU oracle at the small head; more than 256 classes is refused."""
    ffi, synth, HipAZNet, orc = mods
    from aznet_hip.net import HipDetNet
    net, head = small
    fmap = synth.make_feature_map(14, synth.SMALL_DIMS["C"], 38, 63)
    net.set_conv(fmap)
    dhead = synth.make_det_head(seed=7, **dict(synth.SMALL_DET_DIMS, ncls=ncls))
    dnet = HipDetNet(dhead, net)
    rng = np.random.RandomState(ncls)
    x1 = rng.uniform(0, 900, 70); y1 = rng.uniform(0, 500, 70)
    props = np.stack([x1, y1, x1 + rng.uniform(16, 90, 70), y1 + rng.uniform(16, 90, 70)], 1)
    rois = orc.get_rois_blob(props, 1.0)
    p, b = net.ctx.det_forward(rois)
    pr, br = orc.det_head_forward(dhead, fmap[0], rois)
    assert p.shape == (70, ncls) and b.shape == (70, 4 * ncls)
    assert np.abs(p - pr).max() <= 1e-5
    np.testing.assert_allclose(b, br, rtol=1e-4, atol=1e-5)
    s, bx = net.ctx.detect(props, 1.0, 600, 1000)
    sr, bxr = orc.frcnn_forward({"fc": orc.OracleDetNet(dhead)}, (600, 1000), 1.0, props, ncls, {"conv5_3": fmap}, orc.OracleCfg())
    assert np.abs(s - sr).max() <= 1e-5
    np.testing.assert_allclose(bx, bxr, rtol=1e-4, atol=2e-3)
    if ncls == 256:
        with pytest.raises(ffi.AzError):
            HipDetNet(synth.make_det_head(seed=7, **dict(synth.SMALL_DET_DIMS, ncls=257)), net)
        HipDetNet(dhead, net)


def test_channels_last_map_is_borrowed_and_equal(small, mods):
    """A torch.channels_last conv5_3 is read in place (az_set_feature_map_dev_nhwc, no transpose): same proposals
    as the NCHW tensor and as the host array; also through the one-call launch (az_propose_launch_on)."""
    import torch
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    fmap = synth.make_feature_map(12, synth.SMALL_DIMS["C"], 38, 63)
    p = ffi.AzContext.make_params(600, 1000, 1.0, 0.0)
    net.set_conv(fmap)
    want = net.propose(p, want_scores=True)
    t = torch.from_numpy(fmap).cuda()
    tcl = t.contiguous(memory_format=torch.channels_last)
    assert not tcl.is_contiguous()
    for m in (t, tcl):
        net.set_conv(m)
        got = net.propose(p, want_scores=True)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        net.ctx.propose_launch(p, fmap=m)
        got = net.ctx.propose_fetch(want_scores=True)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    from aznet_hip.backbone import VGG16Conv5
    bb = VGG16Conv5(device="cuda:0", seed=2, width_div=32, channels_last_out=True)
    x = torch.zeros(1, 3, 64, 96)
    y = bb(x)
    assert y.is_contiguous(memory_format=torch.channels_last) and tuple(y.shape) == (1, 16, 4, 6)
