"""A stream of DISTINCT images at ONE tuned threshold -- what tools/prop_az.py runs (prop_az.py:74-79: Tz from thresh.pkl;
lib/detect/test.py:508-513: image after image).  The context chooses the form of every search from what the images BEFORE
it looked like, so on a stream its guesses are sometimes wrong (an early end that the tree outgrows, a deferred root whose
level never comes, a pass that carries rows nobody needs): every image of the stream must still come out exactly as the
plain level loop and the CPU oracle give it, whatever was guessed, with searches queued ahead on two lanes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_IMG = 32


@pytest.fixture(scope="module")
def mods():
    from aznet_hip import ffi, synth
    from aznet_hip.net import HipAZNet
    from oracle import az_oracle as orc
    return ffi, synth, HipAZNet, orc


def _plain(ffi, H, W, scale, Tz, **kw):
    return ffi.AzContext.make_params(H, W, scale, Tz, speculate=False, fused=False, fused_levels=False, static_tree=False,
                                     pair_spec=False, full_spec=False, early_end=False, **kw)


def _fmap_hw(synth, H, W, scale):
    return synth.conv_out_size(int(round(H * scale))), synth.conv_out_size(int(round(W * scale)))


def _tune_on_gpu(ffi, net, items, anchors_per_img):
    """detect.tune.tune_thresh's threshold over the set, the scores never leaving HBM (az_tune_*)."""
    net.ctx.tune_begin(len(items) * 2 * net.ctx.max_regions)
    for (H, W, sc, fmap) in items:
        net.set_conv(fmap)
        net.propose(ffi.AzContext.make_params(H, W, sc, 0.0, tune=True))
    tz, npool = net.ctx.tune_kth_largest(len(items) * anchors_per_img)
    net.ctx.tune_end()
    return float(tz), npool


def _tune_on_cpu(orc, head, items, anchors_per_img):
    lists, allz = [], []
    for (H, W, sc, fmap) in items:
        onet = orc.OracleNet(head, feat_fn=lambda d, fmap=fmap: fmap)
        _, Bhis = orc.im_propose_tune({"full": onet, "fc": onet}, (H, W), sc, orc.OracleCfg(Tz=0.0))
        lists.append(Bhis[:, 4])
        allz.append(Bhis[:, 4])
    return float(orc.tune_thresh(lists, len(items) * anchors_per_img)), np.sort(np.concatenate(allz))


def _clear_of_scores(tz, allz, want=2e-5):
    """The tuned threshold IS one of the zoom scores; the oracle's BLAS and the GPU differ by ulps there.  Move it into the
    nearest gap of at least `want` between two scores of the set (the searches on both sides then take the same decisions)."""
    i = int(np.searchsorted(allz, tz))
    for d in range(0, len(allz)):
        for j in (i - d, i + d):
            if 1 <= j < len(allz) and allz[j] - allz[j - 1] >= want:
                return float(0.5 * (allz[j] + allz[j - 1]))
    return tz


def _stream(ffi, net, items, Tz, order, depth):
    import torch
    tmaps = [torch.from_numpy(np.ascontiguousarray(f)).cuda().contiguous(memory_format=torch.channels_last) for (_, _, _, f) in items]
    prm = [ffi.AzContext.make_params(H, W, sc, Tz, static_tree=False) for (H, W, sc, _) in items]
    got, launched = [], 0
    for i in range(len(order)):
        while launched < min(len(order), i + depth):
            k = order[launched]
            net.ctx.propose_launch(prm[k], fmap=tmaps[k], producer_done=True)
            launched += 1
        got.append(net.ctx.propose_fetch(want_scores=True, want_stats=True))
    return got


def _check_set(mods, head, items, anchors_per_img, lanes, depth, expect_forms=None):
    ffi, synth, HipAZNet, orc = mods
    net = HipAZNet(head, name="stream")
    ref = HipAZNet(head, name="stream_ref")
    net.ctx.set_lanes(lanes)
    tz_gpu, npool = _tune_on_gpu(ffi, ref, items, anchors_per_img)
    tz_cpu, allz = _tune_on_cpu(orc, head, items, anchors_per_img)
    assert npool == allz.size
    assert abs(tz_gpu - tz_cpu) <= 1e-4, (tz_gpu, tz_cpu)                 # the tuner itself: same threshold from the same set
    Tz = _clear_of_scores(tz_gpu, allz)
    assert abs(Tz - tz_gpu) < 0.05
    # what every image must give: the plain level loop on the GPU (bit for bit) and the CPU oracle (tolerance)
    want, trees = [], []
    for (H, W, sc, fmap) in items:
        ref.set_conv(fmap)
        Yr, Sr, sr = ref.propose(_plain(ffi, H, W, sc, Tz), want_scores=True, want_stats=True)
        Ya, Sa = ref.ctx.last_candidates()
        onet = orc.OracleNet(head, feat_fn=lambda d, fmap=fmap: fmap)
        Yo, tr = orc.im_propose({"full": onet, "fc": onet}, (H, W), sc, orc.OracleCfg(Tz=Tz), return_trace=True)
        assert [int(sr.level_regions[l]) for l in range(len(tr["levels"]))] == [lv["B"].shape[0] for lv in tr["levels"]]
        assert sr.num_eval == tr["num_eval"] and sr.depth == tr["depth"]
        assert Ya.shape == tr["Y_all"].shape
        assert np.abs(Sa.astype(np.float64) - tr["aScores"]).max() <= 1e-4
        np.testing.assert_allclose(Ya, tr["Y_all"], rtol=1e-4, atol=2e-2)
        want.append((Yr, Sr, sr))
        trees.append(tuple(int(sr.level_regions[l]) for l in range(sr.n_levels)))
    assert len(set(trees)) >= 6, "the set must hold DIFFERENT trees"      # (a stream of one tree would prove nothing)
    # dataset order, twice (the second pass meets the histories the first left), then a shuffled pass
    order = list(range(len(items))) * 2 + [int(i) for i in np.random.RandomState(3).permutation(len(items))]
    got = _stream(ffi, net, items, Tz, order, depth)
    forms, reruns = {}, 0
    for k, (Y, S, st) in zip(order, got):
        Yr, Sr, sr = want[k]
        assert np.array_equal(Y, Yr) and np.array_equal(S, Sr), (k, trees[k], int(st.search_form), int(st.n_reruns))
        assert [int(st.level_regions[l]) for l in range(st.n_levels)] == list(trees[k])
        assert st.num_eval == sr.num_eval and st.depth == sr.depth and st.n_candidates == sr.n_candidates
        assert [int(st.level_unique[l]) for l in range(st.n_levels)] == [int(sr.level_unique[l]) for l in range(sr.n_levels)]
        # the passes of the search cover every level that has regions, each level once
        cover = 0
        for i in range(int(st.n_passes)):
            assert cover & int(st.pass_levels[i]) == 0
            cover |= int(st.pass_levels[i])
        for l in range(st.n_levels):
            if st.level_unique[l] > 0:
                assert (cover >> l) & 1, (l, trees[k], [int(x) for x in st.pass_levels[:st.n_passes]])
        forms[int(st.search_form)] = forms.get(int(st.search_form), 0) + 1
        reruns += int(st.n_reruns)
    return trees, forms, reruns


@pytest.mark.parametrize("lanes,depth", [(2, 3), (1, 2), (2, 4)])
def test_stream_of_object_images_at_the_tuned_threshold(mods, lanes, depth):
    """Deep, sparse, image-dependent trees: planted-object maps and a zoom unit that reads them (synth.make_object_*), the
    threshold tuned at the reference's cfg.TRAIN.ANCHORS_PER_IMG = 20 (config.py:104)."""
    ffi, synth, HipAZNet, orc = mods
    head = synth.make_object_head(seed=77, **synth.SMALL_DIMS)
    C = synth.SMALL_DIMS["C"]
    items = [(600, 1000, 1.0, synth.make_object_map(j, C, 38, 63)) for j in range(N_IMG)]
    trees, forms, reruns = _check_set(mods, head, items, 20, lanes, depth)
    assert max(len([x for x in t if x > 0]) for t in trees) >= 5          # some trees reach the last level ...
    assert min(len([x for x in t if x > 0]) for t in trees) <= 2          # ... and some end with the root's children


@pytest.mark.parametrize("anchors", [20, 600])
def test_stream_of_mixed_shapes_and_densities(mods, anchors):
    """Random weights: the zoom score drifts with region size, so one threshold gives a mixture of trees that end early and
    dense ones; three image shapes interleaved, each with the history of ITS last searches."""
    ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    C = synth.SMALL_DIMS["C"]
    shapes = [(600, 1000, 1.0), (375, 500, 1.6), (480, 640, 1.25)]
    items = []
    for j in range(N_IMG):
        H, W, sc = shapes[j % 3]
        fh, fw = _fmap_hw(synth, H, W, sc)
        items.append((H, W, sc, synth.make_scene_map(j, C, fh, fw)))
    _check_set(mods, head, items, anchors, 2, 3)


@pytest.mark.parametrize("lanes,depth", [(2, 3), (1, 3)])
def test_stream_against_the_reference_run_g14(mods, lanes, depth):
    """g14 (oracle/gen_golden_stream.py): the REFERENCE's own tune_thresh over 12 planted-object images and its own im_propose
    on each at that threshold (lib/detect/tune.py:318-366, lib/detect/test.py:346-414; head on the CPU).  The GPU tuner finds
    the reference's threshold; the stream of the 12 images -- queued, two passes -- forwards the reference's number of unique
    rois at every level of every image and returns its proposals (1e-4-driven tolerances: the reference's head ran on BLAS)."""
    from helpers import load
    ffi, synth, HipAZNet, orc = mods
    g = load("g14_stream.npz")
    n, H, W, Tz = int(g["n_img"]), int(g["H"]), int(g["W"]), float(g["Tz"])
    head = synth.make_object_head(seed=77, **synth.SMALL_DIMS)
    items = [(H, W, 1.0, synth.make_object_map(j, synth.SMALL_DIMS["C"], 38, 63)) for j in range(n)]
    net = HipAZNet(head, name="g14")
    net.ctx.set_lanes(lanes)
    tz_gpu, npool = _tune_on_gpu(ffi, net, items, int(g["anchors_per_img"]))
    assert npool == int(g["pool_size"]) and abs(tz_gpu - float(g["thresh"])) <= 1e-4, (tz_gpu, float(g["thresh"]))
    order = list(range(n)) * 2
    got = _stream(ffi, net, items, Tz, order, depth)
    for k, (Y, S, st) in zip(order, got):
        calls = [int(x) for x in g["calls%d" % k]]
        assert [int(st.level_unique[l]) for l in range(st.n_levels) if st.level_unique[l] > 0] == calls, (k, calls)
        ref = g["Y%d" % k]
        assert Y.shape == ref.shape, (k, Y.shape, ref.shape)
        # every proposal of the reference within reach of one of ours (a tie at the cut may swap the last ones)
        hit = [np.abs(Y - r).max(axis=1).min() <= 1e-3 for r in ref]
        assert np.mean(hit) >= 0.97, (k, float(np.mean(hit)))


@pytest.mark.parametrize("nb", [12, 5])
def test_lockstep_batches_against_the_reference_run_g14(mods, nb):
    """g14 again, the 12 images in lockstep batches (az_batch_launch: every level's rois of a batch in ONE head pass): every
    image forwards the reference's number of unique rois at every level and returns its proposals."""
    import torch
    from helpers import load
    ffi, synth, HipAZNet, orc = mods
    g = load("g14_stream.npz")
    n, H, W, Tz = int(g["n_img"]), int(g["H"]), int(g["W"]), float(g["Tz"])
    head = synth.make_object_head(seed=77, **synth.SMALL_DIMS)
    maps = [torch.from_numpy(synth.make_object_map(j, synth.SMALL_DIMS["C"], 38, 63)).cuda().contiguous(memory_format=torch.channels_last)
            for j in range(n)]
    net = HipAZNet(head, name="g14_batch")
    prm = ffi.AzContext.make_params(H, W, 1.0, Tz)
    got = []
    for rep in range(2):
        got = []
        for i0 in range(0, n, nb):
            net.ctx.batch_launch(prm, maps[i0:i0 + nb], producer_done=True)
            got += net.ctx.batch_fetch_all(want_scores=True, want_stats=True)
    for k, (Y, S, st) in enumerate(got):
        calls = [int(x) for x in g["calls%d" % k]]
        assert st.search_form == 5 and st.n_reruns == 0
        assert [int(st.level_unique[l]) for l in range(st.n_levels) if st.level_unique[l] > 0] == calls, (k, calls)
        ref = g["Y%d" % k]
        assert Y.shape == ref.shape, (k, Y.shape, ref.shape)
        hit = [np.abs(Y - r).max(axis=1).min() <= 1e-3 for r in ref]
        assert np.mean(hit) >= 0.97, (k, float(np.mean(hit)))
