"""Pair speculation in the level loop (az_search.hip: pair_plan, az_level.hip: lookup stage, az_geom_dev.h:
spec_children_rows) vs the plain level loop -- identical bits.

The head pass of level l can also evaluate one row per distinct RoIPool window among ALL children of its regions:
every region of level l+1 is such a child (B(l+1) = _sift_dup(divide_region(B(l)[zoom >= Tz])), test.py:386-390,
div.pyx:15-89), so level l+1's head outputs are looked up instead of computed by a pass of their own.  Everything
observable must equal the search without it -- proposals, scores, the candidate list in the reference's order, the
per-level statistics -- for any Tz, image shape, dedup setting; and the full-size search must still equal the
pure-CPU oracle."""
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
    return HipAZNet(head, name="small_pair"), head


def _scale(H, W):
    scale = 600.0 / min(H, W)
    if np.round(scale * max(H, W)) > 1000:
        scale = 1000.0 / max(H, W)
    return scale


def _run(net, ffi, H, W, scale, Tz, pair, **kw):
    # (full_spec=False: these tests are about the pair rows; the whole-tree pass has tests/test_gpu_full.py)
    Y, S, st = net.propose(ffi.AzContext.make_params(H, W, scale, Tz, static_tree=False, pair_spec=pair, full_spec=False, **kw),
                           want_scores=True, want_stats=True)
    Ya, Sa = net.ctx.last_candidates()
    return dict(Y=Y, S=S, Ya=Ya, Sa=Sa, st=st)


def _same(a, b):
    for k in ("Y", "S", "Ya", "Sa"):
        assert a[k].shape == b[k].shape, k
        assert np.array_equal(a[k], b[k]), k
    sa, sb = a["st"], b["st"]
    for f in ("n_proposals", "num_eval", "depth", "n_levels", "n_candidates"):
        assert getattr(sa, f) == getattr(sb, f), f
    for f in ("level_regions", "level_unique", "level_zoomed"):
        assert list(getattr(sa, f)) == list(getattr(sb, f)), f


def _zoom_quantiles(net, ffi, H, W, scale):
    net.propose(ffi.AzContext.make_params(H, W, scale, 0.0, tune=True))
    return np.sort(net.ctx.last_anchors()[1].astype(np.float64))


@pytest.mark.parametrize("H,W,kw", [
    (600, 1000, {}), (375, 500, {}), (480, 640, {}), (333, 777, {}), (500, 353, {}),
    (600, 1000, {"dedup": 0.0}), (600, 1000, {"min_side": 16}), (600, 1000, {"num_proposals": 50}),
    (600, 1000, {"fixed_num": False, "Tc": 0.4}),
    (800, 1200, {}),               # 6 levels: pairs (4, 5*) and then level 6 outgrows the fused kernel -> rerun
    (200, 90, {}), (1000, 300, {}),
])
def test_pair_speculation_equals_plain_level_loop(small, mods, H, W, kw):
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    scale = _scale(H, W)
    fh, fw = synth.conv_out_size(int(round(H * scale))), synth.conv_out_size(int(round(W * scale)))
    net.set_conv(synth.make_feature_map(41, synth.SMALL_DIMS["C"], fh, fw))
    zs = _zoom_quantiles(net, ffi, H, W, scale)
    tzs = [0.0] + [float(zs[int(q * (len(zs) - 1))]) for q in (0.15, 0.5, 0.8)] + [2.0]
    some_pair = False
    for Tz in tzs:
        a = _run(net, ffi, H, W, scale, Tz, True, **kw)
        b = _run(net, ffi, H, W, scale, Tz, False, **kw)
        _same(a, b)
        # (whether the root's row rides on a later pass is decided from the context's previous search: a tree that ended
        #  early last time keeps it in the first pass, which can save a one-row pass)
        extra = max(0, int(a["st"].root_deferred) - int(b["st"].root_deferred))
        assert b["st"].n_passes + extra >= a["st"].n_passes
        some_pair = some_pair or a["st"].n_passes < b["st"].n_passes
        # ... and whatever the context decides by itself from the searches it has seen
        c = _run(net, ffi, H, W, scale, Tz, None, **kw)
        _same(c, b)
    if min(H, W) >= 320 and (H, W) != (800, 1200):           # (five levels or more: a level behind the first three)
        assert some_pair, "no search of this shape took a pair-speculation pass"


def test_history_turns_pair_speculation_on(small, mods):
    """Without history nothing is speculated; after a search of the same shape the context decides by its cost model
    (az_search.hip: pair_plan): a dense tree (Tz = 0) speculates; whatever it decides for a sparse one, the bits hold."""
    ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    net = HipAZNet(head, name="hist")
    net.ctx.set_pass_costs(ffi.AzContext.REFERENCE_PASS_COSTS)      # (the decision below: not by this box's clock)
    net.set_conv(synth.make_feature_map(41, synth.SMALL_DIMS["C"], 38, 63))
    first = _run(net, ffi, 600, 1000, 1.0, 0.0, None)
    # (49: the root's row rides in the speculative pass -- a context without history does not defer it)
    assert first["st"].n_passes == 3 and list(first["st"].pass_rows[:3])[0] == 49
    second = _run(net, ffi, 600, 1000, 1.0, 0.0, None)
    _same(first, second)
    assert second["st"].n_passes == 2                       # (48 rows) + (level 4 + all children of level 4)
    assert second["st"].pass_rows[1] > 131 + 500
    zs = _zoom_quantiles(net, ffi, 600, 1000, 1.0)
    Tz = float(zs[int(0.9 * (len(zs) - 1))])              # nine regions in ten do not zoom
    sp1 = _run(net, ffi, 600, 1000, 1.0, Tz, None)
    sp2 = _run(net, ffi, 600, 1000, 1.0, Tz, None)
    sp3 = _run(net, ffi, 600, 1000, 1.0, Tz, False)
    _same(sp1, sp3)
    _same(sp2, sp3)
    assert sp2["st"].n_passes <= sp3["st"].n_passes


def test_pair_speculation_full_head_vs_cpu_oracle_and_plain(mods, gemm_mode):
    """Config A at the full head (25088 -> 4096 -> ...): the pair-speculation search equals the plain level loop bit
    for bit (all 8129 candidates) and the pure-CPU oracle within tolerance; a calibrated Tz as well."""
    ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=1234, **synth.FULL_DIMS)
    net = HipAZNet(head, name="full_pair", max_regions=4096, gemm_mode=gemm_mode)
    fmap = synth.make_feature_map(4, 512, 38, 63)
    net.set_conv(fmap)
    onet = orc.OracleNet(head, feat_fn=lambda d: fmap)
    nets = {"full": onet, "fc": onet}
    _, tr0 = orc.im_propose(nets, (600, 1000), 1.0, orc.OracleCfg(Tz=0.0), return_trace=True)
    zs = np.sort(np.concatenate([lv["zoom"][1:] if i == 0 else lv["zoom"] for i, lv in enumerate(tr0["levels"][:3])]))
    tz_c = None
    for j in range(len(zs) // 2, len(zs) - 1):
        if zs[j + 1] - zs[j] > 2e-3:
            tz_c = 0.5 * (zs[j] + zs[j + 1])
            break
    assert tz_c is not None
    for Tz in (0.0, tz_c):
        a = _run(net, ffi, 600, 1000, 1.0, Tz, True)
        b = _run(net, ffi, 600, 1000, 1.0, Tz, False)
        _same(a, b)
        assert a["st"].n_passes == 2 and b["st"].n_passes == 3
        Yref, tr = orc.im_propose(nets, (600, 1000), 1.0, orc.OracleCfg(Tz=Tz), return_trace=True)
        assert a["st"].num_eval == tr["num_eval"] and a["st"].depth == tr["depth"]
        assert a["Ya"].shape == tr["Y_all"].shape
        assert np.abs(a["Sa"].astype(np.float64) - tr["aScores"]).max() <= 1e-4
        np.testing.assert_allclose(a["Ya"], tr["Y_all"], rtol=1e-4, atol=2e-2)
    # the deep tree at the full head: pair (4, 5*), then level 6 (2048 regions) on the multi-launch kernels
    net.set_conv(synth.make_feature_map(5, 512, 38, 57))
    a = _run(net, ffi, 800, 1200, 0.75, 0.0, True)
    b = _run(net, ffi, 800, 1200, 0.75, 0.0, False)
    _same(a, b)


def test_pair_speculation_with_graphs_and_nan_zoom(mods):
    ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=78, **synth.SMALL_DIMS)
    net = HipAZNet(head, name="pair_graph")
    net.set_conv(synth.make_feature_map(43, synth.SMALL_DIMS["C"], 38, 63))
    ref = _run(net, ffi, 600, 1000, 1.0, 0.05, False)
    net.ctx.set_graphs(True)
    try:
        for _ in range(3):
            _same(_run(net, ffi, 600, 1000, 1.0, 0.05, True), ref)
    finally:
        net.ctx.set_graphs(False)
    # NaN zoom scores never pass `zoom >= Tz`: only the forced root divides; the speculative rows are simply unused
    head["bz"] = np.full(1, np.nan, dtype=np.float32)
    net2 = HipAZNet(head, name="pair_nan")
    net2.set_conv(synth.make_feature_map(43, synth.SMALL_DIMS["C"], 38, 63))
    a = _run(net2, ffi, 600, 1000, 1.0, 0.3, True)
    b = _run(net2, ffi, 600, 1000, 1.0, 0.3, False)
    _same(a, b)
    assert [int(a["st"].level_regions[l]) for l in range(5)] == [1, 8, 0, 0, 0]
