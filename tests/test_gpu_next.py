"""GPU parity tests for the rows next to the proposal path, through the C ABI:
image front-end, bbox_overlaps / evaluate_recall, the tuner's search and its threshold select.
f64 / integer work is compared bit-exactly against the goldens the reference's own code produced
(oracle/gen_golden_next.py) and against the oracle on fresh seeded inputs."""
import os
import pickle

import numpy as np
import pytest

from helpers import load

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
    return HipAZNet(head, name="small_next"), head


MEANS = np.array([[[102.9801, 115.9465, 122.7717]]])


# ---------------------------------------------------------------- front-end
@pytest.mark.parametrize("h,w,scale", [(375, 500, 1.6), (600, 1000, 1.0), (333, 500, 600.0 / 333), (500, 1000, 1.0),
                                       (1200, 1600, 0.5), (97, 131, 0.37), (5, 7, 3.0)])
def test_image_blob_bit_exact_vs_oracle(small, mods, h, w, scale):
    ffi, synth, HipAZNet, orc = mods
    ctx = small[0].ctx
    im = synth.make_image(h + w, h, w)
    want = orc.image_blob(im, MEANS, scale)
    got = ctx.image_blob(im, MEANS, scale)
    assert got.shape == want.shape == (1, 3) + orc.image_blob_size(h, w, scale)
    assert np.array_equal(got, want)


def test_image_blob_into_torch_tensor_and_detect_api(small, mods):
    import torch
    ffi, synth, HipAZNet, orc = mods
    ctx = small[0].ctx
    im = synth.make_image(9, 375, 500)
    oh, ow = ctx.image_blob_size(375, 500, 1.6)
    out = torch.empty((1, 3, oh, ow), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    ctx.image_blob(im, MEANS, 1.6, out=out)
    assert np.array_equal(out.cpu().numpy(), orc.image_blob(im, MEANS, 1.6))
    from detect import test as T
    from detect.config import cfg
    cfg.TEST.MAX_SIZE = 1000
    blob, scales = T._get_image_blob(im)
    assert scales[0] == 1.6 and np.array_equal(blob, orc.image_blob(im, cfg.PIXEL_MEANS, 1.6))
    with pytest.raises(ffi.AzError):                 # NULL image
        ctx._chk(ctx.L.az_image_blob_host(ctx.h, None, 10, 10, None, 1.0, None, 10, 10))


# ---------------------------------------------------------------- recall evaluation
def test_bbox_overlaps_golden_and_random(small, mods):
    ffi, synth, HipAZNet, orc = mods
    ctx = small[0].ctx
    g = load("g6_bbox_overlaps.npz")
    assert np.array_equal(ctx.bbox_overlaps(g["boxes"], g["query"]), g["overlaps"])
    rng = np.random.RandomState(4)
    for n, k in ((1, 1), (300, 7), (2000, 33)):
        a = rng.uniform(0, 400, (n, 4)); a[:, 2:] += a[:, :2]
        q = np.floor(rng.uniform(0, 400, (k, 4))); q[:, 2:] += q[:, :2]
        a[: min(n, k)] = q[: min(n, k)]                       # exact matches -> IoU 1
        assert np.array_equal(ctx.bbox_overlaps(a, q), orc.bbox_overlaps(a, q))
    assert ctx.bbox_overlaps(np.zeros((0, 4)), np.zeros((3, 4))).shape == (0, 3)
    import utils.cython_bbox as cb
    assert np.array_equal(cb.bbox_overlaps(g["boxes"], g["query"]), g["overlaps"])
    with pytest.raises(ValueError):
        cb.bbox_overlaps(g["boxes"].astype(np.float32), g["query"])


def test_evaluate_recall_golden(small, mods):
    ffi, synth, HipAZNet, orc = mods
    from datasets.imdb import imdb
    g = load("g10_recall.npz")
    n = int(g["n_img"])

    class Db(imdb):
        def gt_roidb(self):
            return [{"boxes": g["gt%d" % i], "gt_classes": g["cls%d" % i]} for i in range(n)]

    db = Db("golden")
    db._image_index = list(range(n))
    ar, gt_overlaps, recalls, thresholds = db.evaluate_recall([g["cand%d" % i] for i in range(n)], ctx=small[0].ctx)
    assert np.array_equal(gt_overlaps, g["gt_overlaps"])
    assert np.array_equal(recalls, g["recalls"]) and np.array_equal(thresholds, g["thresholds"])
    assert ar == float(g["ar"])
    # the reference asserts when an image has more gt boxes than candidates
    with pytest.raises(AssertionError):
        db.evaluate_recall(_too_few(g, n), ctx=small[0].ctx)


def _too_few(g, n):
    """Candidate lists where one image with >= 2 gt boxes gets a single candidate."""
    out = [np.zeros((0, 4)) for _ in range(n)]
    for i in range(n):
        if (g["cls%d" % i] > 0).sum() >= 2:
            out[i] = g["cand%d" % i][:1]
            return out
    raise RuntimeError("fixture has no image with two gt boxes")


def test_recall_match_many_images_vs_oracle(small, mods):
    ffi, synth, HipAZNet, orc = mods
    rng = np.random.RandomState(12)
    cands, gts = [], []
    for i in range(200):
        k = int(rng.randint(0, 9))
        n = int(rng.randint(max(k, 1), 320))
        gt = np.floor(rng.uniform(0, 300, (k, 4))); gt[:, 2:] += gt[:, :2] + 5
        b = rng.uniform(0, 300, (n, 4)); b[:, 2:] += b[:, :2] + 5
        b[:k] = gt + rng.uniform(-8, 8, (k, 4))
        if k and i % 3 == 0:
            b[-1] = gt[0]
        cands.append(b)
        gts.append(gt)
    got = small[0].ctx.recall_match(cands, gts)
    assert np.array_equal(got, orc.recall_gt_overlaps(cands, gts))


# ---------------------------------------------------------------- tuner
def _injected(net, fmap):
    class Injected(object):
        name = "inj"
        blobs = net.blobs

        def forward(self, blobs=None, **kw):
            kw.pop("data", None)
            kw["conv5_3"] = fmap
            return net.forward(blobs=blobs, **kw)
    return Injected()


@pytest.mark.parametrize("H,W,tzq,batch", [(375, 500, None, 10000), (480, 640, None, 10000), (375, 500, 0.6, 10000),
                                           (600, 1000, 0.5, 300), (100, 64, None, 10000)])
def test_tuner_search_vs_oracle_loop(small, mods, H, W, tzq, batch):
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    scale = 600.0 / min(H, W)
    if np.round(scale * max(H, W)) > 1000:
        scale = 1000.0 / max(H, W)
    fh, fw = synth.conv_out_size(int(round(H * scale))), synth.conv_out_size(int(round(W * scale)))
    fmap = synth.make_feature_map(6, synth.SMALL_DIMS["C"], fh, fw)
    net.set_conv(fmap)
    nprop = 2000                                                      # cfg.TRAIN.NUM_PROPOSALS
    Tz = 0.0
    if tzq is not None:
        net.propose(ffi.AzContext.make_params(H, W, scale, 0.0, num_proposals=nprop, batch_size=batch, tune=True))
        z = net.ctx.last_anchors()[1].astype(np.float64)
        Tz = float(np.quantile(z, tzq))
    params = ffi.AzContext.make_params(H, W, scale, Tz, num_proposals=nprop, batch_size=batch, tune=True)
    Y, S, st = net.propose(params, want_scores=True, want_stats=True)
    regions, zoom = net.ctx.last_anchors()
    inj = _injected(net, fmap)
    cfg = orc.OracleCfg(Tz=Tz, BATCH_SIZE=batch, NUM_PROPOSALS=nprop)
    Y5, Bhis = orc.im_propose_tune({"full": inj, "fc": inj}, (H, W), scale, cfg)
    assert np.array_equal(regions, Bhis[:, :4])                       # anchor regions: bit-exact f64
    assert np.array_equal(zoom.astype(np.float64), Bhis[:, 4])        # zoom scores: same bits (same head)
    assert st.num_eval == Bhis.shape[0]
    assert Y.shape == Y5[:, :4].shape
    assert np.array_equal(np.sort(S.astype(np.float64)), np.sort(Y5[:, 4]))
    # the search of lib/detect/test.py on the same image walks one level less
    st2 = net.propose(ffi.AzContext.make_params(H, W, scale, Tz, batch_size=batch), want_stats=True)[1]
    assert st.n_levels == st2.n_levels + 1
    with pytest.raises(ffi.AzError):
        net.ctx.last_anchors()                                        # last search was not a tuner search


def test_tune_threshold_select_golden(small, mods):
    ctx = small[0].ctx
    for per_img in (20, 400):
        g = load("g12_tune_thresh_%d.npz" % per_img)
        lists = [g["bhis%d" % i][:, -1].astype(np.float32) for i in range(3)]
        ctx.tune_begin(sum(a.size for a in lists))
        for a in lists:
            ctx.tune_push(a)
        v, n = ctx.tune_kth_largest(3 * per_img)
        assert n == sum(a.size for a in lists) and np.float64(v) == float(g["thresh"])
        top = ctx.tune_top(3 * per_img)
        assert top.size >= 3 * per_img and top.min() == np.float32(v)
        v2, _ = ctx.tune_kth_largest(10 ** 7)
        assert v2 == float("-inf")
        ctx.tune_end()


def test_tune_kth_largest_random_with_ties(small, mods):
    ffi, synth, HipAZNet, orc = mods
    ctx = small[0].ctx
    rng = np.random.RandomState(8)
    a = rng.uniform(0, 1, 300000).astype(np.float32)
    a[::7] = a[3]                                   # heavy ties
    a[5] = -0.25; a[6] = 0.0                        # a negative value and a zero
    ctx.tune_begin(a.size)
    ctx.tune_push(a[:100000]); ctx.tune_push(a[100000:])
    s = np.sort(a)[::-1]
    for k in (1, 2, 1000, 42857, 299999):
        assert ctx.tune_kth_largest(k)[0] == s[k - 1]
    assert ctx.tune_kth_largest(a.size)[0] == float("-inf")
    with pytest.raises(ffi.AzError):
        ctx.tune_push(np.zeros(10, np.float32))     # pool is full
    ctx.tune_end()


def test_tune_thresh_end_to_end_vs_heap(small, mods, tmp_path):
    """detect.tune.tune_thresh over a synthetic imdb == the reference's heap over the same
    per-image anchor scores (orc.tune_thresh)."""
    ffi, synth, HipAZNet, orc = mods
    net, head = small
    from detect import tune as U
    from detect.config import cfg, cfg_set_mode, cfg_set_path
    import detect.config as C
    from datasets.factory import get_imdb
    old_tz, old_np = cfg.SEAR.get("Tz", 0.0), cfg.SEAR.get("NUM_PROPOSALS", 300)
    cfg_set_mode("Train")
    cfg_set_path("pytest_tune")
    old_root = cfg.ROOT_DIR
    cfg.ROOT_DIR = str(tmp_path)
    cfg.TEST.MAX_SIZE = 1000
    try:
        db = get_imdb("synthetic_120x200_5")
        fmaps = {}

        class FakeBackbone(object):                 # conv5_3 as a function of the image blob's size
            device = "cuda:0"

            def __call__(self, blob):
                import torch
                key = tuple(blob.shape[2:])
                if key not in fmaps:
                    fmaps[key] = synth.make_feature_map(40 + len(fmaps), synth.SMALL_DIMS["C"],
                                                        synth.conv_out_size(key[0]), synth.conv_out_size(key[1]))
                return torch.from_numpy(fmaps[key]).to("cuda:0")

        net.backbone = FakeBackbone()
        for per_img in (3, 50, 10 ** 5):
            cfg.TRAIN.ANCHORS_PER_IMG = per_img
            got = U.tune_thresh({"full": net, "fc": net}, db)
            # the multi-GPU form: two "ranks" tune disjoint shards, rank 0 merges the per-rank top scores
            db.shard = [1, 3]
            ctx = net.ctx
            ctx.tune_begin(2 * 2 * ctx.max_regions)
            for i in db.shard:
                U._search(net, db.image_at(i))
            other_top = ctx.tune_top(5 * per_img)
            ctx.tune_end()
            db.shard = [0, 2, 4]
            merged = U.tune_thresh({"full": net, "fc": net}, db, gather=lambda top: [top, other_top])
            db.shard = None
            assert merged == got
            lists = []
            for i in range(5):
                _, Bhis = U.im_propose(net, db.image_at(i))
                lists.append(Bhis[:, 4])
            want = orc.tune_thresh(lists, 5 * per_img)
            assert got == want
            with open(os.path.join(C.get_output_dir(db, net), "thresh.pkl"), "rb") as f:
                assert pickle.load(f) == want
    finally:
        net.backbone = None
        cfg.ROOT_DIR = old_root
        cfg.TRAIN.ANCHORS_PER_IMG = 20
        cfg.SEAR.Tz, cfg.SEAR.NUM_PROPOSALS = old_tz, old_np
        cfg_set_path(None)
