"""Tz <= 0: all levels in one head pass (az_static.hip) vs the level loop -- identical bits.

lib/detect/test.py:383-390 selects `indZ = np.where(zoom >= Tz)`; a Sigmoid output is never below 0, so for
Tz <= 0 (config.py:275, the TRAIN-phase setting) the tree is a function of the image shape.  az_propose then
forwards the rois of every level at once.  Everything observable must equal the level-by-level search:
proposals, scores, the candidate list in the reference's order, and the per-level statistics.  The premise is
checked on the device: a NaN zoom score (which the reference's comparison rejects) sends the image back to
the level loop."""
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
def small(mods):
    ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    return HipAZNet(head, name="small"), head


def _scale(H, W):
    scale = 600.0 / min(H, W)
    if np.round(scale * max(H, W)) > 1000:
        scale = 1000.0 / max(H, W)
    return scale


def _both(net, ffi, H, W, scale, Tz, **kw):
    outs = []
    for static in (True, False):
        Y, S, st = net.propose(ffi.AzContext.make_params(H, W, scale, Tz, static_tree=static, **kw),
                               want_scores=True, want_stats=True)
        Ya, Sa = net.ctx.last_candidates()
        outs.append(dict(Y=Y, S=S, Ya=Ya, Sa=Sa, st=st))
    return outs


def _same(a, b, nlev_check=True):
    for k in ("Y", "S", "Ya", "Sa"):
        assert a[k].shape == b[k].shape, k
        assert np.array_equal(a[k], b[k]), k
    sa, sb = a["st"], b["st"]
    for f in ("n_proposals", "num_eval", "depth", "n_levels", "n_candidates"):
        assert getattr(sa, f) == getattr(sb, f), f
    for f in ("level_regions", "level_unique", "level_zoomed"):
        assert list(getattr(sa, f)) == list(getattr(sb, f)), f


@pytest.mark.parametrize("H,W,Tz,kw", [
    (600, 1000, 0.0, {}),
    (600, 1000, -0.25, {}),
    (600, 1000, 0.0, {"num_proposals": 50}),
    (600, 1000, 0.0, {"fixed_num": False, "Tc": 0.5}),
    (600, 1000, 0.0, {"dedup": 0.0}),                  # cfg.DEDUP_BOXES <= 0: every region forwarded
    (600, 1000, 0.0, {"min_side": 16}),
    (480, 640, 0.0, {"batch_size": 100}),              # dedup in chunks of BATCH_SIZE regions
    (375, 500, 0.0, {}),
    (800, 1200, 0.0, {}),                              # BASELINE config 4: K = 7, 2729 regions
    (200, 90, 0.0, {}), (60, 1000, 0.0, {}), (1000, 40, 0.0, {}), (21, 21, 0.0, {}), (333, 777, 0.0, {}),
])
def test_static_plan_equals_level_loop(small, mods, H, W, Tz, kw):
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    scale = _scale(H, W)
    fh, fw = synth.conv_out_size(int(round(H * scale))), synth.conv_out_size(int(round(W * scale)))
    net.set_conv(synth.make_feature_map(11, synth.SMALL_DIMS["C"], fh, fw))
    a, b = _both(net, ffi, H, W, scale, Tz, **kw)
    assert a["st"].static_plan == 1 and b["st"].static_plan == 0
    assert a["st"].spec_rows == sum(a["st"].level_unique[l] for l in range(a["st"].n_levels))
    _same(a, b)
    # every region of every level zoomed
    assert list(a["st"].level_zoomed) == list(a["st"].level_regions)


def test_static_plan_is_per_shape_and_survives_shape_changes(small, mods):
    """The plan is cached per image shape / scale / MIN_SIDE / dedup / batch: alternate between shapes and settings."""
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    seq = [(600, 1000, {}), (375, 500, {}), (600, 1000, {}), (600, 1000, {"min_side": 16}), (600, 1000, {}),
           (600, 1000, {"dedup": 0.0}), (600, 1000, {"batch_size": 7}), (600, 1000, {})]
    ref = {}
    for H, W, kw in seq:
        scale = _scale(H, W)
        fh, fw = synth.conv_out_size(int(round(H * scale))), synth.conv_out_size(int(round(W * scale)))
        net.set_conv(synth.make_feature_map(12, synth.SMALL_DIMS["C"], fh, fw))
        key = (H, W, tuple(sorted(kw.items())))
        Y, S, st = net.propose(ffi.AzContext.make_params(H, W, scale, 0.0, **kw), want_scores=True, want_stats=True)
        assert st.static_plan == 1
        if key not in ref:
            Yl, Sl = net.propose(ffi.AzContext.make_params(H, W, scale, 0.0, static_tree=False, **kw), want_scores=True)
            ref[key] = (Yl, Sl)
        assert np.array_equal(Y, ref[key][0]) and np.array_equal(S, ref[key][1])


def test_positive_tz_never_takes_the_static_plan(small, mods):
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    net.set_conv(synth.make_feature_map(11, synth.SMALL_DIMS["C"], 38, 63))
    for Tz in (1e-12, 0.3):
        Y, st = net.propose(ffi.AzContext.make_params(600, 1000, 1.0, Tz), want_stats=True)
        assert st.static_plan == 0
    # the tuner's variant never does either (it records the anchor history level by level)
    net.propose(ffi.AzContext.make_params(600, 1000, 1.0, 0.0, tune=True))
    assert net.ctx.last_anchors()[0].shape[0] > 0


def test_nan_zoom_scores_fall_back_to_the_level_loop(mods):
    """`zoom >= Tz` is False for NaN (test.py:386): with a NaN zoom bias only the forced root divides.  The static
    plan's premise check catches it on the device and the image is rerun level by level."""
    ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=78, **synth.SMALL_DIMS)
    head["bz"] = np.full(1, np.nan, dtype=np.float32)
    net = HipAZNet(head, name="nanzoom")
    fmap = synth.make_feature_map(13, synth.SMALL_DIMS["C"], 38, 63)
    net.set_conv(fmap)
    a, b = _both(net, ffi, 600, 1000, 1.0, 0.0)
    assert a["st"].static_plan == 0                       # (the answer came from the rerun)
    _same(a, b)
    assert [int(a["st"].level_regions[l]) for l in range(5)] == [1, 8, 0, 0, 0]
    # the oracle's loop agrees on the tree
    onet = orc.OracleNet(head, feat_fn=lambda d: fmap)
    with np.errstate(invalid="ignore"):
        _, tr = orc.im_propose({"full": onet, "fc": onet}, (600, 1000), 1.0, orc.OracleCfg(Tz=0.0), return_trace=True)
    assert tr["num_eval"] == 9 and a["st"].num_eval == 9
    # and a later healthy search on the same context takes the plan again
    net2 = HipAZNet(synth.make_head(seed=77, **synth.SMALL_DIMS), name="ok")
    net2.set_conv(fmap)
    assert net2.propose(ffi.AzContext.make_params(600, 1000, 1.0, 0.0), want_stats=True)[1].static_plan == 1


def test_static_plan_with_graphs_and_staged_result(small, mods):
    """hipGraph replay and the device-resident result record work on the one-pass plan too."""
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    net.set_conv(synth.make_feature_map(11, synth.SMALL_DIMS["C"], 38, 63))
    p = ffi.AzContext.make_params(600, 1000, 1.0, 0.0)
    Y0, S0 = net.propose(p, want_scores=True)
    net.ctx.set_graphs(True)
    try:
        for _ in range(3):
            Y1, S1 = net.propose(p, want_scores=True)
            assert np.array_equal(Y0, Y1) and np.array_equal(S0, S1)
    finally:
        net.ctx.set_graphs(False)


def test_full_head_one_pass_bits_equal_level_loop(mods, gemm_mode):
    """Config A at the full head: the one-pass search (int6 through the 12-wave many-row GEMM of az_head12.hip) and the
    level loop (k_fc_splitk at 48 / 131 / 517 rows) give the same bits -- proposals, scores, all 8129 candidates."""
    ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=1234, **synth.FULL_DIMS)
    net = HipAZNet(head, name="full", max_regions=4096, gemm_mode=gemm_mode)
    net.set_conv(synth.make_feature_map(4, 512, 38, 63))
    a, b = _both(net, ffi, 600, 1000, 1.0, 0.0)
    assert a["st"].static_plan == 1 and a["st"].spec_rows == 688 and b["st"].static_plan == 0
    assert a["Ya"].shape[0] > 8000
    _same(a, b)
    # the level loop again: the context now expects many rows at level 5 (517 rois in the search just fetched) and
    # sends that level's int6 to both GEMM kernels, of which the many-row one owns the launch -- same bits
    p_ll = ffi.AzContext.make_params(600, 1000, 1.0, 0.0, static_tree=False)
    Y2, S2, st2 = net.propose(p_ll, want_scores=True, want_stats=True)
    Ya2, Sa2 = net.ctx.last_candidates()
    _same(dict(Y=Y2, S=S2, Ya=Ya2, Sa=Sa2, st=st2), b)
    # ... and at BASELINE config 4's tree (2672 rois in one pass)
    net.set_conv(synth.make_feature_map(5, 512, 38, 57))
    a, b = _both(net, ffi, 800, 1200, 0.75, 0.0)
    assert a["st"].static_plan == 1 and b["st"].static_plan == 0
    _same(a, b)


def test_plan_cache_eviction_with_graphs(mods, monkeypatch):
    """The per-shape plans live in an LRU cache (AZ_PLAN_CACHE entries, default 64): cycle five shapes through a
    two-entry cache, with hipGraph replay on (a dropped plan takes the captured launch sequences with it)."""
    ffi, synth, HipAZNet, orc = mods
    monkeypatch.setenv("AZ_PLAN_CACHE", "2")
    net = HipAZNet(synth.make_head(seed=77, **synth.SMALL_DIMS), name="lru")
    shapes = [(600, 1000), (375, 500), (480, 640), (333, 777), (500, 353)]
    ref = {}
    net.ctx.set_graphs(True)
    try:
        for rnd in range(3):
            for H, W in shapes:
                scale = _scale(H, W)
                fh, fw = synth.conv_out_size(int(round(H * scale))), synth.conv_out_size(int(round(W * scale)))
                net.set_conv(synth.make_feature_map(21, synth.SMALL_DIMS["C"], fh, fw))
                Y, S, st = net.propose(ffi.AzContext.make_params(H, W, scale, 0.0), want_scores=True, want_stats=True)
                assert st.static_plan == 1
                if (H, W) not in ref:
                    net.ctx.set_graphs(False)
                    ref[(H, W)] = net.propose(ffi.AzContext.make_params(H, W, scale, 0.0, static_tree=False), want_scores=True)
                    net.ctx.set_graphs(True)
                assert np.array_equal(Y, ref[(H, W)][0]) and np.array_equal(S, ref[(H, W)][1])
    finally:
        net.ctx.set_graphs(False)
