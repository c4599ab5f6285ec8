"""Properties of the path that hold at ANY size, checked at BASELINE.json's full sizes (600x1000 and the deep
800x1200 tree, 8129 / 30 019 candidates) where the CPU oracle takes too long to be the checker: sortedness,
idempotence, exact inverses, self-consistency of a search's outputs.  Every call goes through the C ABI; the reference
formulas restated here are one-liners over NumPy (hash keys, f32 IoU) -- the properties, not a second implementation.

  _sift_dup   lib/utils/div.pyx:78-89      output ascending and unique in the 10-px hash; a second pass changes nothing
  dedup       lib/detect/test.py:210-218   rois[index][inv_index] reproduces every roi's hash; unique rois dedup to themselves
  nms         lib/utils/nms.pyx:17-68      keep in descending score order; kept boxes pairwise below the threshold; every
                                           dropped box overlaps a better kept one; nms(kept) keeps everything
  top-K       lib/detect/test.py:393-401   the 300 proposals ARE the 300 best of the search's own candidate list
  RoIPool     test_fc.prototxt:14-25       a roi covering the map pools to the map's channel maxima
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from aznet_hip import ffi, synth
    from aznet_hip.net import HipAZNet
    return ffi, synth, HipAZNet


@pytest.fixture(scope="module")
def full(mods):
    ffi, synth, HipAZNet = mods
    head = synth.make_head(seed=1234, **synth.FULL_DIMS)
    return HipAZNet(head, name="full_props", max_regions=4096), head


@pytest.fixture(scope="module")
def geo(mods):
    """a context with room for the deep tree's last division (2048 parents -> 8192 children); the head is irrelevant here"""
    ffi, synth, HipAZNet = mods
    return HipAZNet(synth.make_head(seed=77, **synth.SMALL_DIMS), name="geo_props", max_regions=16384)


def _sift_key(B, side=10.0):
    """the hash _sift_dup sorts by (div.pyx:80-82): np.round is half-to-even, the products are exact integers < 2^53"""
    return np.round(B / side).dot(np.array([1.0, 1e3, 1e6, 1e9]))


def _tree(ctx, H, W, levels):
    B = np.array([[0.0, 0.0, W - 1.0, H - 1.0]])
    out = [B]
    for _ in range(levels):
        B = ctx.divide_region(B, 10.0)
        out.append(B)
    return out


@pytest.mark.parametrize("H,W,levels", [(600, 1000, 5), (800, 1200, 6)])
def test_divide_region_output_is_sorted_unique_and_a_fixed_point_of_sift_dup(geo, H, W, levels):
    net = geo
    tree = _tree(net.ctx, H, W, levels)
    sizes = [len(b) for b in tree]
    if (H, W) == (600, 1000):
        assert sizes == [1, 8, 32, 134, 564, 2253]           # the full tree of config A (+ one level)
    else:
        assert sizes[:6] == [1, 8, 32, 128, 512, 2048]       # config 4
    for B in tree[1:]:
        k = _sift_key(B)
        assert np.all(np.diff(k) > 0)                        # ascending, no duplicate hash
        assert np.array_equal(net.ctx.sift_dup(B, 10.0), B)  # idempotent
        # children stay inside the root up to the +1 px per level the reference's cells grow by (div.pyx:47-58)
        assert B[:, 0].min() >= 0 and B[:, 1].min() >= 0
        assert B[:, 2].max() <= W - 1 + levels and B[:, 3].max() <= H - 1 + levels
        assert np.all(B[:, 2] > B[:, 0]) and np.all(B[:, 3] > B[:, 1])


@pytest.mark.parametrize("scale", [1.0, 0.75, 1.6])
def test_roi_dedup_is_an_exact_inverse_and_idempotent(geo, scale):
    net = geo
    B = _tree(net.ctx, 800, 1200, 6)[-1]                     # 8k+ regions
    rois, index, inv = net.ctx.roi_dedup(B, scale, 1.0 / 16.0, 10000)
    assert rois.shape == (len(B), 5) and inv.shape == (len(B),)
    key = np.round(rois * np.float32(0.0625)).astype(np.float64).dot(np.array([1.0, 1e3, 1e6, 1e9, 1e12]))
    ku = key[index]
    assert np.all(np.diff(ku) > 0)                           # np.unique's order: ascending hash, each once
    assert np.array_equal(ku[inv], key)                      # scatter-back reproduces every roi's hash
    first = np.full(len(ku), len(B), dtype=np.int64)
    np.minimum.at(first, inv, np.arange(len(B)))
    assert np.array_equal(first, index)                      # return_index: FIRST occurrence
    # the unique rois' boxes are their own dedup
    Bu = B[index]
    _, index2, inv2 = net.ctx.roi_dedup(Bu, scale, 1.0 / 16.0, 10000)
    assert len(index2) == len(Bu) and np.array_equal(np.sort(index2), np.arange(len(Bu)))
    assert np.array_equal(index2[inv2], np.arange(len(Bu)))


def _iou_f32(a, b):
    """nms.pyx:52-65 in float32, one box against many"""
    one = np.float32(1.0)
    xx1 = np.maximum(a[0], b[:, 0]); yy1 = np.maximum(a[1], b[:, 1])
    xx2 = np.minimum(a[2], b[:, 2]); yy2 = np.minimum(a[3], b[:, 3])
    w = np.maximum(np.float32(0.0), xx2 - xx1 + one); h = np.maximum(np.float32(0.0), yy2 - yy1 + one)
    inter = w * h
    area_a = (a[2] - a[0] + one) * (a[3] - a[1] + one)
    area_b = (b[:, 2] - b[:, 0] + one) * (b[:, 3] - b[:, 1] + one)
    return inter / (area_a + area_b - inter)


@pytest.mark.parametrize("N,thresh", [(8129, 0.5), (8129, 0.7), (30019, 0.5), (2000, 0.3)])
def test_nms_properties_at_proposal_scale(geo, N, thresh):
    net = geo
    rng = np.random.RandomState(N + int(thresh * 10))
    x1 = rng.uniform(0, 900, N); y1 = rng.uniform(0, 500, N)
    dets = np.stack([x1, y1, x1 + rng.uniform(10, 210, N), y1 + rng.uniform(10, 210, N),
                     rng.permutation(N) / float(N)], 1).astype(np.float32)        # distinct scores
    keep = np.asarray(net.ctx.nms(dets, thresh))
    assert len(set(keep.tolist())) == len(keep) and keep.min() >= 0 and keep.max() < N
    sc = dets[keep, 4]
    assert np.all(np.diff(sc) < 0)                           # descending score order
    assert keep[0] == int(np.argmax(dets[:, 4]))             # the best box is always kept
    K = dets[keep]
    kept = np.zeros(N, dtype=bool); kept[keep] = True
    th = np.float64(thresh)
    # kept boxes: pairwise below the threshold (every kept box against the better kept ones)
    step = max(1, len(K) // 400)                             # a few hundred rows of the kept x kept matrix
    for i in range(1, len(K), step):
        assert np.all(_iou_f32(K[i], K[:i]).astype(np.float64) < th), i
    # dropped boxes: each overlaps a kept box with a higher score at >= thresh (checked on a sample)
    dropped = np.flatnonzero(~kept)
    for j in dropped[:: max(1, len(dropped) // 300)]:
        better = K[K[:, 4] > dets[j, 4]]
        assert len(better) and np.any(_iou_f32(dets[j], better).astype(np.float64) >= th), j
    # idempotence: nothing more to suppress among the kept boxes, same order
    again = np.asarray(net.ctx.nms(K, thresh))
    assert np.array_equal(again, np.arange(len(K)))


@pytest.mark.parametrize("H,W,tzq", [(600, 1000, None), (600, 1000, 0.3), (800, 1200, None)])
def test_search_outputs_are_self_consistent_at_full_size(full, mods, H, W, tzq):
    """One search at BASELINE's sizes: the proposals are the best of its own candidate list, in order, inside the image;
    the statistics add up; a second run gives the same bits."""
    ffi, synth, HipAZNet = mods
    net, head = full
    fh, fw = (H + 15) // 16, (W + 15) // 16
    fmap = synth.make_feature_map(11, 512, fh, fw)
    net.set_conv(fmap)
    scale = 600.0 / min(H, W)
    Tz = 0.0
    if tzq is not None:
        # a threshold inside the zoom scores of the tree's upper levels: a pruned, data-dependent tree
        B = _tree(net.ctx, H, W, 2)[-1]
        rois = np.hstack([np.zeros((len(B), 1)), B * scale]).astype(np.float32)
        z, _, _ = net.ctx.head_forward(rois)
        Tz = float(np.quantile(z, tzq))
    prm = ffi.AzContext.make_params(H, W, scale, Tz)
    Y, S, st = net.propose(prm, want_scores=True, want_stats=True)
    Yall, Sall = net.ctx.last_candidates()
    n = st.n_levels
    assert st.num_eval == sum(st.level_regions[:n]) and st.n_candidates == len(Sall) == len(Yall)
    assert all(st.level_unique[l] <= st.level_regions[l] for l in range(n))
    assert all(st.level_zoomed[l] <= st.level_regions[l] for l in range(n))
    if tzq is None and (H, W) == (600, 1000):
        # (739 regions x 11 sub-regions = 8129 candidates before the MIN_SIDE filter, test.py:171-187, drops the clipped slivers)
        assert list(st.level_regions[:5]) == [1, 8, 32, 134, 564] and 8000 < len(Sall) <= 8129
    if tzq is None and (H, W) == (800, 1200):
        assert st.num_eval == 2729
    k = min(300, len(Sall))
    assert Y.shape == (k, 4) and np.all(np.diff(S) <= 0)
    order = np.argsort(-Sall.astype(np.float64), kind="stable")[:k]
    assert np.array_equal(S, Sall[order])                    # the k best scores of the candidate list ...
    ties = np.concatenate([[False], np.diff(S) == 0]) | np.concatenate([np.diff(S) == 0, [False]])
    assert np.array_equal(Y[~ties], Yall[order][~ties])      # ... with their boxes (tied scores: argsort(-aScores) is unstable)
    assert np.all(Yall[:, 0] >= 0) and np.all(Yall[:, 1] >= 0) and np.all(Yall[:, 2] <= W - 1) and np.all(Yall[:, 3] <= H - 1)
    assert np.all(np.minimum(Yall[:, 2] - Yall[:, 0], Yall[:, 3] - Yall[:, 1]) + 1 >= 10)
    Y2, S2 = net.propose(prm, want_scores=True)
    assert np.array_equal(Y, Y2) and np.array_equal(S, S2)


def test_roi_pool_of_the_whole_map_is_the_channel_maximum(full, mods):
    ffi, synth, HipAZNet = mods
    net, _ = full
    fmap = synth.make_feature_map(5, 512, 38, 63)
    net.set_conv(fmap)
    # a roi that covers the whole map (and overhangs it: clamped), one that is a single cell
    rois = np.array([[0, 0, 0, 1007, 607], [0, -50, -50, 2000, 2000], [0, 160, 320, 160, 320]], dtype=np.float32)
    p = net.ctx.roi_pool(rois).reshape(3, 512, 49)
    cmax = fmap[0].reshape(512, -1).max(axis=1)
    assert np.array_equal(p[0].max(axis=1), cmax)
    assert np.array_equal(p[1].max(axis=1), cmax)
    assert np.array_equal(p[2], np.repeat(fmap[0][:, 20, 10][:, None], 49, axis=1))      # round(320/16) = 20, round(160/16) = 10


def test_bbox_overlaps_is_symmetric_with_a_unit_diagonal(geo):
    """bbox.pyx:132-172 at 3000 x 3000: IoU(A, B) = IoU(B, A)^T bit for bit (the f64 operations commute), IoU(A, A) has ones
    on the diagonal, everything lies in [0, 1], disjoint boxes give exactly 0."""
    net = geo
    rng = np.random.RandomState(7)
    N = 3000
    x1 = rng.uniform(0, 900, N); y1 = rng.uniform(0, 500, N)
    A = np.stack([x1, y1, x1 + rng.uniform(10, 210, N), y1 + rng.uniform(10, 210, N)], 1)
    x1 = rng.uniform(0, 900, N); y1 = rng.uniform(0, 500, N)
    B = np.stack([x1, y1, x1 + rng.uniform(10, 210, N), y1 + rng.uniform(10, 210, N)], 1)
    ab = net.ctx.bbox_overlaps(A, B)
    ba = net.ctx.bbox_overlaps(B, A)
    assert ab.shape == (N, N) and np.array_equal(ab, ba.T)
    aa = net.ctx.bbox_overlaps(A, A)
    assert np.array_equal(np.diag(aa), np.ones(N)) and np.array_equal(aa, aa.T)
    assert ab.min() >= 0.0 and ab.max() <= 1.0
    far = A + np.array([5000.0, 5000.0, 5000.0, 5000.0])
    assert not net.ctx.bbox_overlaps(A[:500], far[:500]).any()
