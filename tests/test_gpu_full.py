"""Whole-tree speculation in the level loop (az_search.hip: full_prepare, az_static.hip: window table, az_fused.hip /
az_level.hip: lookup stages) vs the plain level loop -- identical bits.

For a dense tree the search's ONE head pass evaluates the unique rois of the image shape's FULL tree (the one-pass plan's
rows, plus the speculative rows whose window the plan lacks); every level then finds its regions' head outputs by
RoIPool window (a roi's outputs are a function of its pooled window only) and decodes them against its own boxes.
Valid for any Tz: a pruned tree may keep another _sift_dup survivor than the full tree (same 10-px hash, other
coordinates, other window); a search that needs a window the pass did not evaluate is repeated level by level.
The CLOSURE rows (full_spec="closure") hold one row per distinct window among ALL regions any pruning can produce
(level l+1 = every child of every region of level l, no _sift_dup): that pass serves every Tz and is never repeated.
Everything observable must equal the search without it, and the full-size search must equal the pure-CPU oracle."""
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
    net = HipAZNet(head, name="small_full")
    # the form a search takes "by history" is decided from head-pass costs the context measures on its device; the tests
    # that assert a form pin the table (the full head's figures) so that they do not depend on the box or the head size
    net.ctx.set_pass_costs(ffi.AzContext.REFERENCE_PASS_COSTS)
    return net, head


def _scale(H, W):
    scale = 600.0 / min(H, W)
    if np.round(scale * max(H, W)) > 1000:
        scale = 1000.0 / max(H, W)
    return scale


def _run(net, ffi, H, W, scale, Tz, full, **kw):
    Y, S, st = net.propose(ffi.AzContext.make_params(H, W, scale, Tz, static_tree=False, full_spec=full, **kw),
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


def _zooms(net, ffi, H, W, scale):
    net.propose(ffi.AzContext.make_params(H, W, scale, 0.0, tune=True))
    return np.sort(net.ctx.last_anchors()[1].astype(np.float64))


SHAPES = [(600, 1000), (375, 500), (480, 640), (500, 353), (333, 500), (720, 1280), (600, 600), (420, 1000)]


@pytest.mark.parametrize("H,W", SHAPES, ids=["%dx%d" % s for s in SHAPES])
def test_forced_whole_tree_pass_equals_plain_level_loop(small, mods, H, W):
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    scale = _scale(H, W)
    fh, fw = synth.conv_out_size(int(round(H * scale))), synth.conv_out_size(int(round(W * scale)))
    if orc.num_levels(H, W) - 1 < 4:
        pytest.skip("fewer than four levels: nothing behind the speculative ones")
    for seed in (5, 6):
        net.set_conv(synth.make_feature_map(seed, synth.SMALL_DIMS["C"], fh, fw))
        z = _zooms(net, ffi, H, W, scale)
        for Tz in (0.0, float(np.quantile(z, 0.3)), float(np.quantile(z, 0.6)), float(z[len(z) // 2]), 1.5):
            for kw in ({}, {"dedup": 0.0}, {"num_proposals": 2000}):
                plain = _run(net, ffi, H, W, scale, Tz, False, pair_spec=False, **kw)
                full = _run(net, ffi, H, W, scale, Tz, True, **kw)
                _same(plain, full)
                # (a level of more than 1024 regions outgrows the fused level kernel: such trees stay level by level)
                # (so do shapes whose speculative pass has more than 64 rows)
                if Tz == 0.0 and not kw and (H, W) in ((600, 1000), (375, 500), (480, 640), (600, 600)):
                    assert full["st"].n_passes == 1, (full["st"].n_passes, list(full["st"].pass_rows[:4]))


ONE_PASS_SHAPES = ((600, 1000), (375, 500), (480, 640), (600, 600))


@pytest.mark.parametrize("H,W", SHAPES, ids=["%dx%d" % s for s in SHAPES])
def test_forced_closure_pass_equals_plain_level_loop_and_is_never_repeated(small, mods, H, W):
    """The closure rows serve EVERY Tz: same bits as the plain level loop, one head pass, no search run twice."""
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    scale = _scale(H, W)
    fh, fw = synth.conv_out_size(int(round(H * scale))), synth.conv_out_size(int(round(W * scale)))
    if orc.num_levels(H, W) - 1 < 4:
        pytest.skip("fewer than four levels: nothing behind the speculative ones")
    rows = set()
    for seed in (5, 6):
        net.set_conv(synth.make_feature_map(seed, synth.SMALL_DIMS["C"], fh, fw))
        z = _zooms(net, ffi, H, W, scale)
        for Tz in (0.0, float(np.quantile(z, 0.1)), float(np.quantile(z, 0.3)), float(np.quantile(z, 0.6)),
                   float(z[len(z) // 2]), 1.5):
            for kw in ({}, {"dedup": 0.0}, {"num_proposals": 2000}):
                plain = _run(net, ffi, H, W, scale, Tz, False, pair_spec=False, **kw)
                clos = _run(net, ffi, H, W, scale, Tz, "closure", **kw)
                _same(plain, clos)
                assert clos["st"].n_reruns == 0 or clos["st"].search_form != 3
                if (H, W) in ONE_PASS_SHAPES and not kw:
                    st = clos["st"]
                    assert st.search_form == 3 and st.n_passes == 1 and st.n_reruns == 0, \
                        (Tz, st.search_form, st.n_passes, st.n_reruns, list(st.pass_rows[:4]))
                    rows.add(int(st.pass_rows[0]))
    if (H, W) in ONE_PASS_SHAPES:
        assert len(rows) == 1                           # a property of the image shape, whatever the Tz
        full = _run(net, ffi, H, W, scale, 0.0, True)
        assert rows.pop() >= int(full["st"].pass_rows[0])       # the closure holds the full tree's rows


def test_tree_rows_can_miss_a_window_that_the_closure_holds(small, mods):
    """Why the closure exists: a pruned tree may keep a _sift_dup survivor the full tree drops (other coordinates, other
    RoIPool window).  The pass over the full tree's rows then lacks a window and the search is run again level by level
    (n_reruns = 1, same bits); the closure pass of the same search is not."""
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    H, W = 600, 1000
    missed = 0
    for seed in (5, 6, 7, 8):
        net.set_conv(synth.make_feature_map(seed, synth.SMALL_DIMS["C"], 38, 63))
        z = _zooms(net, ffi, H, W, 1.0)
        for q in (0.02, 0.05, 0.1, 0.15, 0.2, 0.3, 0.4, 0.5):
            Tz = float(np.quantile(z, q))
            plain = _run(net, ffi, H, W, 1.0, Tz, False, pair_spec=False)
            tree = _run(net, ffi, H, W, 1.0, Tz, True)
            clos = _run(net, ffi, H, W, 1.0, Tz, "closure")
            _same(plain, tree)
            _same(plain, clos)
            assert clos["st"].n_reruns == 0 and clos["st"].n_passes == 1 and clos["st"].search_form == 3
            assert tree["st"].n_reruns in (0, 1)
            missed += int(tree["st"].n_reruns)
    assert missed > 0, "no pruned tree of these 32 needed a window the full tree's rows lack"


def test_history_takes_the_closure_for_dense_pruned_trees_when_it_is_cheaper(mods):
    """By history and row counts: after a pruned tree the candidate is the closure pass, taken when one pass of its rows
    is cheaper than the passes the tree's rows would cost level by level -- here with a pinned cost table whose rows are
    cheap (as with int6 on the 16-bit matrix cores), so that a dense pruned tree takes it."""
    ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    net = HipAZNet(head, name="hist_closure")
    net.ctx.set_pass_costs(((40, 140.0), (704, 300.0)))
    assert net.ctx.pass_costs() == ((40, 140.0), (704, 300.0))
    H, W = 600, 1000
    net.set_conv(synth.make_feature_map(9, synth.SMALL_DIMS["C"], 38, 63))
    z = _zooms(net, ffi, H, W, 1.0)
    pd = ffi.AzContext.make_params(H, W, 1.0, float(np.quantile(z, 0.1)), static_tree=False)
    plain = net.propose(ffi.AzContext.make_params(H, W, 1.0, float(np.quantile(z, 0.1)), static_tree=False, full_spec=False,
                                                  pair_spec=False), want_scores=True, want_stats=True)
    a = net.propose(pd, want_scores=True, want_stats=True)
    b = net.propose(pd, want_scores=True, want_stats=True)
    assert plain[2].level_zoomed[2] < plain[2].level_regions[2] or plain[2].level_zoomed[3] < plain[2].level_regions[3]
    assert b[2].search_form == 3 and b[2].n_passes == 1 and b[2].n_reruns == 0, (b[2].search_form, b[2].n_passes)
    for r in (a, b):
        assert np.array_equal(r[0], plain[0]) and np.array_equal(r[1], plain[1])
    # ... and with rows at the fp32-MFMA price the same tree stays level by level (the closure's extra rows cost more
    # than the pass they save)
    net.ctx.set_pass_costs(ffi.AzContext.REFERENCE_PASS_COSTS)
    c = net.propose(pd, want_scores=True, want_stats=True)
    assert c[2].search_form in (0, 1) and np.array_equal(c[0], plain[0])


def test_history_turns_it_on_for_dense_trees_only(mods):
    """The tree-rows pass costs a second search when it lacks a window, so one full tree is not enough to take it: the
    shape's last TWO searches must both have walked the full tree (a stream of different images at a tuned Tz rarely does
    that; a context at Tz <= 0 always does).  A pruned history goes back to the level-by-level forms."""
    ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    net = HipAZNet(head, name="hist_dense")
    net.ctx.set_pass_costs(ffi.AzContext.REFERENCE_PASS_COSTS)
    H, W = 600, 1000
    net.set_conv(synth.make_feature_map(9, synth.SMALL_DIMS["C"], 38, 63))
    p0 = ffi.AzContext.make_params(H, W, 1.0, 0.0, static_tree=False)
    a = net.propose(p0, want_scores=True, want_stats=True)          # first search of the shape: no history
    b = net.propose(p0, want_scores=True, want_stats=True)          # one full tree seen: not yet
    c3 = net.propose(p0, want_scores=True, want_stats=True)         # two in a row: one pass over the full tree's rows
    assert a[2].n_passes >= 2 and b[2].search_form != 2 and c3[2].n_passes == 1 and c3[2].search_form == 2
    for r in (b, c3):
        assert np.array_equal(a[0], r[0]) and np.array_equal(a[1], r[1])
    z = _zooms(net, ffi, H, W, 1.0)
    psparse = ffi.AzContext.make_params(H, W, 1.0, float(np.quantile(z, 0.6)), static_tree=False)
    for _ in range(4):
        net.propose(psparse)
    c = net.propose(psparse, want_stats=True)                        # pruned trees seen: level by level again
    assert c[1].num_eval < c3[2].num_eval and c[1].pass_rows[0] < c3[2].pass_rows[0] and c[1].search_form in (0, 1)
    d = net.propose(p0, want_scores=True, want_stats=True)          # (history says pruned: the full tree is found out first)
    e = net.propose(p0, want_scores=True, want_stats=True)
    f = net.propose(p0, want_scores=True, want_stats=True)
    assert d[2].search_form != 2 and f[2].n_passes == 1 and f[2].search_form == 2
    for r in (d, e, f):
        assert np.array_equal(r[0], a[0]) and np.array_equal(r[1], a[1])


def test_full_head_whole_tree_pass_vs_pure_cpu_oracle(mods, gemm_mode):
    ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=1234, **synth.FULL_DIMS)
    net = HipAZNet(head, name="full_whole", max_regions=4096, gemm_mode=gemm_mode)
    H, W = 600, 1000
    fmap = synth.make_feature_map(31, 512, 38, 63)
    net.set_conv(fmap)
    onet = orc.OracleNet(head, feat_fn=lambda d: fmap)
    nets = {"full": onet, "fc": onet}
    _, tr0 = orc.im_propose(nets, (H, W), 1.0, orc.OracleCfg(Tz=0.0), return_trace=True)
    zs = np.sort(np.concatenate([lv["zoom"] for lv in tr0["levels"][1:3]]))
    j = next(j for j in range(len(zs) // 4, len(zs) - 1) if zs[j + 1] - zs[j] > 2e-3)
    for Tz in (0.0, 0.5 * (zs[j] + zs[j + 1])):
        Yref, tr = orc.im_propose(nets, (H, W), 1.0, orc.OracleCfg(Tz=Tz), return_trace=True)
        z = np.concatenate([lv["zoom"] for lv in tr["levels"]])
        assert np.abs(z - Tz).min() > 2e-4
        for form in (True, "closure"):
            Y, S, st = net.propose(ffi.AzContext.make_params(H, W, 1.0, Tz, static_tree=False, full_spec=form),
                                   want_scores=True, want_stats=True)
            # (the pruned tree may need a window the full tree's rows lack: then the search was repeated level by level;
            #  the closure rows -- 773 at 600x1000 against the full tree's 688 -- serve every tree)
            assert st.n_passes == 1 or (Tz > 0.0 and form is True)
            if form == "closure":
                assert st.search_form == 3 and st.n_reruns == 0 and st.n_passes == 1
                assert 688 < st.pass_rows[0] < 900, st.pass_rows[0]
            assert st.depth == tr["depth"] and st.num_eval == tr["num_eval"]
            for l, lev in enumerate(tr["levels"]):
                assert st.level_regions[l] == lev["B"].shape[0]
                assert st.level_unique[l] == sum(f["U"] for f in lev["fwd"])
                assert st.level_zoomed[l] == len(lev["indZ"])
            Yall, Sall = net.ctx.last_candidates()
            assert Yall.shape == tr["Y_all"].shape
            assert np.abs(Sall.astype(np.float64) - tr["aScores"]).max() <= 1e-4
            np.testing.assert_allclose(Yall, tr["Y_all"], rtol=1e-4, atol=2e-2)


def test_deep_tree_takes_the_whole_tree_pass_too(small, mods):
    """BASELINE config 4's shape (800x1200 at 0.75: six levels, 2048 regions at the last): the levels that outgrow the
    fused level kernel run on the multi-launch geometry kernels and find their outputs with the chip-wide lookup."""
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    H, W, scale = 800, 1200, 0.75
    fh, fw = synth.conv_out_size(int(round(H * scale))), synth.conv_out_size(int(round(W * scale)))
    net.set_conv(synth.make_feature_map(17, synth.SMALL_DIMS["C"], fh, fw))
    plain = _run(net, ffi, H, W, scale, 0.0, False, pair_spec=False)
    assert [int(plain["st"].level_regions[l]) for l in range(6)] == [1, 8, 32, 128, 512, 2048]
    _run(net, ffi, H, W, scale, 0.0, True)          # (the first fused attempt of the shape learns which level overflows)
    full = _run(net, ffi, H, W, scale, 0.0, True)
    _same(plain, full)
    assert full["st"].n_passes == 1 and full["st"].pass_rows[0] >= 2672
    z = _zooms(net, ffi, H, W, scale)
    for q in (0.2, 0.5):
        Tz = float(np.quantile(z, q))
        base = _run(net, ffi, H, W, scale, Tz, False, pair_spec=False)
        _same(base, _run(net, ffi, H, W, scale, Tz, True))
        clos = _run(net, ffi, H, W, scale, Tz, "closure")
        _same(base, clos)
        assert clos["st"].n_reruns == 0 or clos["st"].search_form != 3


def test_history_is_kept_per_image_shape(small, mods):
    """A dataset mixes image shapes: each shape keeps the history of its own last searches, so alternating shapes still
    reach the speculative forms from their third search on."""
    ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    net = HipAZNet(head, name="mixed")
    net.ctx.set_pass_costs(ffi.AzContext.REFERENCE_PASS_COSTS)
    shapes = [(600, 1000, 1.0), (375, 500, 1.6), (480, 640, 1.25)]
    maps = [synth.make_feature_map(50 + i, synth.SMALL_DIMS["C"], synth.conv_out_size(int(round(H * sc))),
                                   synth.conv_out_size(int(round(W * sc)))) for i, (H, W, sc) in enumerate(shapes)]
    first, passes = {}, {}
    for rnd in range(4):
        for i, (H, W, sc) in enumerate(shapes):
            net.set_conv(maps[i])
            Y, S, st = net.propose(ffi.AzContext.make_params(H, W, sc, 0.0, static_tree=False), want_scores=True,
                                   want_stats=True)
            passes.setdefault(i, []).append(int(st.n_passes))
            if rnd == 0:
                first[i] = (Y, S)
            else:
                assert np.array_equal(Y, first[i][0]) and np.array_equal(S, first[i][1])
    for i in range(len(shapes)):
        # (the whole-tree pass over the full tree's rows from the third search of a shape on: two full trees in a row)
        assert passes[i][0] >= 2 and passes[i][2] == 1 and passes[i][3] == 1, passes
