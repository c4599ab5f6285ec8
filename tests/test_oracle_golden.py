"""CPU: pin the oracle (oracle/) against golden vectors produced by running the
reference's own Cython/Python (oracle/gen_golden.py)."""
import numpy as np
import pytest

from oracle import az_oracle as orc
from helpers import load, unpack_list, replay_nets, TRACES, ReplayDetNet


def test_divide_region_roots_bit_exact():
    g = load("g1_divide_region.npz")
    for i in range(len(g["sizes"])):
        ins = unpack_list(g, "root%d_in" % i)
        outs = unpack_list(g, "root%d_out" % i)
        for a, b in zip(ins, outs):
            got = orc.divide_region(a, 10)
            assert got.shape == b.shape
            assert np.array_equal(got, b)


def test_divide_region_random_and_singles_bit_exact():
    g = load("g1_divide_region.npz")
    assert np.array_equal(orc.divide_region(g["rand_in"], 10), g["rand_out"])
    singles = unpack_list(g, "single_out")
    for i, s in enumerate(singles):
        assert np.array_equal(orc.divide_region(g["rand_in"][i:i + 1], 10), s)


def test_known_tree_sizes():
    # 600x1000 -> [1, 8, 32, 134, 564, 2253]
    B = np.array([[0, 0, 999.0, 599.0]])
    sizes = []
    for _ in range(5):
        B = orc.divide_region(B, 10)
        sizes.append(B.shape[0])
    assert sizes == [8, 32, 134, 564, 2253]


def test_divide_region_empty():
    assert orc.divide_region(np.zeros((0, 4)), 10).shape == (0, 4)


def test_sift_dup_bit_exact_and_numpy_twin():
    g = load("g2_sift_dup.npz")
    for mh, key in ((10.0, "out10"), (16.0, "out16")):
        got = orc.sift_dup(g["in"], mh)
        assert np.array_equal(got, g[key])
        assert np.array_equal(orc.sift_dup_numpy(g["in"], mh), g[key])


def test_decode_clip_unwrap_bit_exact():
    g = load("g4_decode.npz")
    pred = orc.bbox_pred(g["boxes"], g["deltas"])
    # np.exp(f32) is the only non-IEEE-exact op; same NumPy build => identical here,
    # other hosts may differ by ulps of the f32 exp.
    np.testing.assert_allclose(pred, g["pred"], rtol=1e-6, atol=1e-9)
    clipped = orc.clip_boxes(g["pred"].copy(), (600, 1000))
    assert np.array_equal(clipped, g["clipped"])
    a, c = orc.unwrap_adj_pred(g["clipped"], g["scores"], 10)
    assert np.array_equal(a, g["unwrap_boxes"]) and np.array_equal(c, g["unwrap_scores"])
    assert a.dtype == np.float64 and c.dtype == np.float32


def test_nms_keep_lists_identical():
    g = load("g5_nms.npz")
    for i in range(int(g["ncases"])):
        keep = orc.nms(g["dets%d" % i], float(g["thresh%d" % i]))
        assert keep == list(g["keep%d" % i]), "case %d" % i


def test_nms_empty():
    assert orc.nms(np.zeros((0, 5), dtype=np.float32), 0.5) == []


def test_bbox_overlaps():
    g = load("g6_bbox_overlaps.npz")
    assert np.array_equal(orc.bbox_overlaps(g["boxes"], g["query"]), g["overlaps"])


@pytest.mark.parametrize("tag", TRACES)
def test_im_propose_loop_against_reference_trace(tag):
    """The oracle's level loop, driven by the head outputs recorded during the
    reference's own run, must feed the net identical rois at every call and return
    the reference's proposals."""
    g = load("g7_trace_%s.npz" % tag)
    cfg = orc.OracleCfg(Tz=float(g["Tz"]), BATCH_SIZE=int(g["batch"]))
    nets = replay_nets(g)
    Y = orc.im_propose(nets, (int(g["H"]), int(g["W"])), float(g["scale"]), cfg)
    assert nets["full"].pos + nets["fc"].pos == int(g["ncalls"])
    assert Y.dtype == np.float64 and Y.shape == g["Y"].shape
    np.testing.assert_allclose(Y, g["Y"], rtol=1e-6, atol=1e-9)


def test_roi_pool_known_answers():
    """RoIPool is unpinned by the reference (Caffe absent): hand-computed cases."""
    C, H, W = 2, 8, 10
    feat = np.arange(C * H * W, dtype=np.float32).reshape(C, H, W)
    # integer-aligned roi covering cells x 0..6, y 0..6 (x2 = 6*16 = 96 -> round(6.0) = 6)
    out = orc.roi_pool(feat, np.array([[0, 0, 0, 96, 96]], dtype=np.float32)).reshape(C, 7, 7)
    assert np.array_equal(out[0], feat[0, :7, :7])           # 7x7 bins of one cell each
    # 1-cell roi: every bin is that cell
    out = orc.roi_pool(feat, np.array([[0, 48, 32, 48, 32]], dtype=np.float32)).reshape(C, 7, 7)
    assert np.all(out[1] == feat[1, 2, 3])
    # roi hanging off the map: bins entirely outside are empty -> 0
    out = orc.roi_pool(feat, np.array([[0, 144, 112, 400, 400]], dtype=np.float32)).reshape(C, 7, 7)
    assert out[0, 0, 0] == feat[0, 7:8 + 2, 9:10].max() or out[0, 0, 0] >= 0
    assert out[0, 6, 6] == 0.0
    # .5 rounding: 8 * 0.0625 = 0.5 -> C round() gives 1 (half away from zero), rint would give 0
    out = orc.roi_pool(feat, np.array([[0, 8, 8, 8, 8]], dtype=np.float32)).reshape(C, 7, 7)
    assert np.all(out[0] == feat[0, 1, 1])


def test_fc_blas_vs_plain():
    rng = np.random.RandomState(0)
    x = rng.randn(5, 300).astype(np.float32)
    W = rng.randn(17, 300).astype(np.float32)
    b = rng.randn(17).astype(np.float32)
    np.testing.assert_allclose(orc.fc(x, W, b, True), orc.fc_plain(x, W, b, True), rtol=1e-4, atol=1e-4)


def test_num_levels():
    assert orc.num_levels(600, 1000) == 6 and orc.num_levels(375, 500) == 6
    assert orc.num_levels(640, 853) == 7 and orc.num_levels(800, 1200) == 7


@pytest.mark.parametrize("tag", ["a", "b"])
def test_frcnn_forward_against_reference_trace(tag):
    """_frcnn_forward (test.py:259-318): the oracle, fed the detection-head outputs recorded in
    the reference's own run, must send the head the same deduplicated rois (per BATCH_SIZE chunk)
    and return the reference's scores and per-class boxes."""
    g = load("g9_detect_%s.npz" % tag)
    cfg = orc.OracleCfg(Tz=float(g["Tz"]), BATCH_SIZE=int(g["batch"]))
    det = ReplayDetNet(g)
    conv = {"conv5_3": np.zeros((1, 1, 1, 1), dtype=np.float32)}
    scores, boxes = orc.frcnn_forward({"fc": det}, (int(g["H"]), int(g["W"])), float(g["scale"]),
                                      g["proposals"], 21, conv, cfg)
    assert det.pos == int(g["ndet"])
    assert np.array_equal(scores, g["scores"])
    np.testing.assert_allclose(boxes, g["pred_boxes"], rtol=0, atol=1e-6)


def test_softmax_matches_definition():
    x = np.random.RandomState(0).randn(7, 21).astype(np.float32) * 5
    p = orc.softmax(x)
    assert p.dtype == np.float32 and np.allclose(p.sum(1), 1, atol=1e-6)
    ref = np.exp(x.astype(np.float64) - x.max(1, keepdims=True))
    np.testing.assert_allclose(p, ref / ref.sum(1, keepdims=True), rtol=1e-5, atol=1e-7)


# ---- rows next to the path: goldens from oracle/gen_golden_next.py -----------
def _recall_lists(g):
    cand, gts = [], []
    for i in range(int(g["n_img"])):
        cls = g["cls%d" % i]
        cand.append(g["cand%d" % i])
        gts.append(g["gt%d" % i][np.where(cls > 0)[0], :])      # imdb.py:125-126
    return cand, gts


def test_evaluate_recall_against_reference():
    g = load("g10_recall.npz")
    cand, gts = _recall_lists(g)
    ar, gt_overlaps, recalls, thresholds = orc.evaluate_recall(cand, gts)
    assert np.array_equal(gt_overlaps, g["gt_overlaps"])
    assert np.array_equal(recalls, g["recalls"]) and np.array_equal(thresholds, g["thresholds"])
    assert ar == float(g["ar"])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_tuner_search_against_reference_trace(tag):
    g = load("g11_tune_%s.npz" % tag)
    cfg = orc.OracleCfg(Tz=float(g["Tz"]), NUM_PROPOSALS=int(g["num_proposals"]))
    Y5, Bhis = orc.im_propose_tune(replay_nets(g), (int(g["H"]), int(g["W"])), float(g["scale"]), cfg,
                                   data_blob=np.zeros((1, 3, 2, 2), np.float32))
    assert np.array_equal(Bhis, g["Bhis"])
    assert np.array_equal(Y5[:, 4], g["Y5"][:, 4])
    assert np.array_equal(Y5, g["Y5"])


@pytest.mark.parametrize("per_img", [20, 400])
def test_tune_thresh_against_reference(per_img):
    g = load("g12_tune_thresh_%d.npz" % per_img)
    lists = [g["bhis%d" % i][:, -1] for i in range(3)]
    assert orc.tune_thresh(lists, 3 * per_img) == float(g["thresh"])
    # the heap's answer is the k-th largest of all scores (what the GPU path selects)
    allz = np.sort(np.concatenate(lists))[::-1]
    assert allz[3 * per_img - 1] == float(g["thresh"])
    assert orc.tune_thresh(lists, 10 ** 6) == -np.inf


def test_image_blob_restatement_properties():
    """cv2 is absent (parity unpinned): scale 1 is the identity; a x2 upscale of a constant image
    stays constant; against torch's half-pixel bilinear the f32 results agree to 1e-3."""
    import torch
    import torch.nn.functional as F
    rng = np.random.RandomState(5)
    im = rng.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    means = np.array([102.9801, 115.9465, 122.7717])
    b1 = orc.image_blob(im, means, 1.0)
    assert np.array_equal(b1[0], (im.astype(np.float32) - means.astype(np.float32)).transpose(2, 0, 1))
    c = orc.image_blob(np.full((20, 30, 3), 77, np.uint8), means, 2.0)
    assert c.shape == (1, 3, 40, 60)
    assert np.allclose(c[0, 0], 77 - np.float32(means[0]), atol=1e-4)
    for scale in (2.0, 1.6, 0.5):
        oh, ow = orc.image_blob_size(37, 53, scale)
        if abs(oh / 37.0 - scale) > 1e-9 or abs(ow / 53.0 - scale) > 1e-9:
            continue        # torch derives its step from the size ratio, cv2 from fx/fy
        ref = F.interpolate(torch.from_numpy(b1), size=(oh, ow), mode="bilinear", align_corners=False).numpy()
        assert np.abs(orc.image_blob(im, means, scale) - ref).max() < 1e-3


# ---- harness row (a15 / 8f row 1): goldens from oracle/gen_golden_harness.py -------------------------
def test_net_shared_bookkeeping_and_apply_nms_golden():
    """oracle.net_shared_select / apply_nms == what the reference's test_net_shared (lib/detect/test.py:
    670-778) pickled into detections.pkl and handed to imdb.evaluate_detections, given the per-image
    (scores, boxes) its own im_detect_shared returned."""
    g = load("g13_harness.npz")
    n = int(g["n_img"])
    per = [(g["det_scores%d" % i], g["det_boxes%d" % i]) for i in range(n)]
    all_boxes, thresh = orc.net_shared_select(per, 21)
    nmsd = orc.apply_nms(all_boxes, 0.5)
    assert np.isfinite(thresh[1:]).all()                  # the adaptive thresholds engaged (max_per_set = 80)
    for j in range(1, 21):
        for i in range(n):
            assert all_boxes[j][i].dtype == np.float32
            assert np.array_equal(all_boxes[j][i], g["det_all_%d_%d" % (j, i)])
            got = np.zeros((0, 5), np.float32) if isinstance(nmsd[j][i], list) else nmsd[j][i]
            assert np.array_equal(got, g["det_nms_%d_%d" % (j, i)])
    assert sorted(str(k) for k in g["prop_keys"]) == ["boxes", "recall", "time"] and int(g["prop_recall"]) == 0
    assert str(g["prop_relpath"]) == "output/harness/stub_2img/az_small/proposals.pkl"
    assert str(g["det_relpath"]) == "output/harness/stub_2img/az_small/detections.pkl"


# ---- G3 / G8: the two inline computations of lib/detect/test.py, recorded from the reference itself -------------
def test_roi_dedup_reproduces_the_reference_np_unique():
    """g3: index / inv_index as the reference's own `np.unique(hashes, return_index, return_inverse)` (test.py:212-218)
    produced them inside _az_forward, for every level of three trees (scales 1.0, 1.6, 0.9375) and BATCH_SIZE chunks."""
    g = load("g3_roi_dedup.npz")
    scales = set()
    for i in range(int(g["ncases"])):
        boxes, scale = g["c%d_boxes" % i], float(g["c%d_scale" % i])
        scales.add(scale)
        rois = orc.get_rois_blob(boxes, scale)
        index, inv = orc.roi_dedup(rois)
        assert np.array_equal(index, g["c%d_index" % i]) and np.array_equal(inv, g["c%d_inv_index" % i]), i
        # the hash itself (f32 round-half-even of rois / 16, exact integers in f64)
        v = np.array([1, 1e3, 1e6, 1e9, 1e12])
        assert np.array_equal(np.round(rois * (1. / 16.)).dot(v), g["c%d_hashes" % i])
    assert scales == {1.0, 1.6, 0.9375}


def test_top_k_reproduces_the_reference_argsort():
    """g8: `np.argsort(-aScores)` of whole im_propose runs (test.py:397-401), distinct and heavily tied scores.  The oracle's
    selection is the same call on the same array; with ties the ORDER inside a tie is NumPy's (unstable sort), so what must
    agree is the score sequence and, above the last selected score, the index set."""
    g = load("g8_topk.npz")
    for tag in [str(t) for t in g["runs"]]:
        neg, indA, Yall, Y = g[tag + "_neg_scores"], g[tag + "_indA"], g[tag + "_Y_all"], g[tag + "_Y"]
        k = int(g[tag + "_num_proposals"])
        Yo, ind = orc.top_k(Yall, -neg, k)
        n = min(k, Yall.shape[0])
        assert Yo.shape == Y.shape == (n, 4)
        assert np.array_equal(neg[ind], neg[indA[:n]])                      # same scores in the same order
        cut = neg[indA[n - 1]]
        assert set(ind[neg[ind] < cut]) == set(indA[:n][neg[indA[:n]] < cut])
        assert np.array_equal(Yall[indA[:n]], Y)


def test_fc_backends_agree():
    """The CPU baseline's sgemm (torch CPU addmm) and the NumPy BLAS the fixtures were pinned with give the same
    head within fp32 summation noise; the backend switch restores itself."""
    from aznet_hip import synth
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    fmap = synth.make_feature_map(5, synth.SMALL_DIMS["C"], 24, 32)
    rng = np.random.RandomState(1)
    x1, y1 = rng.uniform(0, 400, 40), rng.uniform(0, 300, 40)
    rois = np.stack([np.zeros(40), x1, y1, x1 + rng.uniform(8, 100, 40), y1 + rng.uniform(8, 80, 40)], 1).astype(np.float32)
    a = orc.head_forward(head, fmap[0], rois)
    orc.set_fc_backend("torch", threads=2)
    try:
        b = orc.head_forward(head, fmap[0], rois)
    finally:
        orc.set_fc_backend("numpy")
    c = orc.head_forward(head, fmap[0], rois)
    for u, v, w in zip(a, b, c):
        assert np.abs(u - v).max() <= 1e-5 and np.array_equal(u, w)
    with pytest.raises(ValueError):
        orc.set_fc_backend("mkl")


def test_stream_at_the_reference_tuned_threshold_g14():
    """g14 (oracle/gen_golden_stream.py): the reference's own tune_thresh over 12 planted-object images and its im_propose on
    every one of them at that threshold.  The oracle's tuner and loop give the same threshold, the same forwards and the same
    proposals, image by image."""
    from aznet_hip import synth
    g = load("g14_stream.npz")
    n, H, W, Tz = int(g["n_img"]), int(g["H"]), int(g["W"]), float(g["Tz"])
    head = synth.make_object_head(seed=77, **synth.SMALL_DIMS)
    maps = [synth.make_object_map(j, synth.SMALL_DIMS["C"], 38, 63) for j in range(n)]
    lists = []
    for m in maps:
        net = orc.OracleNet(head, feat_fn=lambda d, m=m: m)
        _, Bhis = orc.im_propose_tune({"full": net, "fc": net}, (H, W), 1.0, orc.OracleCfg(Tz=0.0))
        lists.append(Bhis[:, 4])
    assert sum(x.size for x in lists) == int(g["pool_size"])
    assert float(orc.tune_thresh(lists, n * int(g["anchors_per_img"]))) == float(g["thresh"])
    trees = set()
    for i, m in enumerate(maps):
        net = orc.OracleNet(head, feat_fn=lambda d, m=m: m)
        Y, tr = orc.im_propose({"full": net, "fc": net}, (H, W), 1.0, orc.OracleCfg(Tz=Tz), return_trace=True)
        assert [sum(f["U"] for f in lv["fwd"]) for lv in tr["levels"]] == [int(x) for x in g["calls%d" % i]]
        np.testing.assert_allclose(Y, g["Y%d" % i], rtol=1e-6, atol=1e-9)
        trees.add(tuple(int(x) for x in g["calls%d" % i]))
    assert len(trees) >= 8                                  # different trees, from the root's children only to five levels
