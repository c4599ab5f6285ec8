"""NumPy/C restatement of AZ-Net's proposal search -- TEST INFRASTRUCTURE ONLY.

Nothing under ``az-net_amd/`` may import this module.  It is the checker used
by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``; it is never the thing shipped or the thing whose speed is the
headline number.

Every function cites the reference file:line it follows (paths relative to the
upstream az-net tree).  The geometry / loop functions are pinned by golden
vectors produced by running the reference's own Python + Cython in the build
container (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``).  The head
arithmetic (RoIPool, InnerProduct, Sigmoid) lives in the reference's absent
``caffe-fast-rcnn`` submodule and is therefore PARITY UNPINNED: it restates
the published Fast R-CNN Caffe layers named by
``models/Pascal/VGG16/az-net/test_fc.prototxt:14-232``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    """Compile oracle/c/az_oracle.c with plain gcc (see oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "c", "az_oracle.c"))):
        subprocess.check_call(["make", "-s", "-C", _HERE, "_build/liboracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        c_dp = ctypes.POINTER(ctypes.c_double)
        c_fp = ctypes.POINTER(ctypes.c_float)
        c_lp = ctypes.POINTER(ctypes.c_long)
        L.orc_divide_children.restype = ctypes.c_long
        L.orc_divide_children.argtypes = [c_dp, ctypes.c_long, c_dp, ctypes.c_long]
        L.orc_sift_dup.restype = ctypes.c_long
        L.orc_sift_dup.argtypes = [c_dp, ctypes.c_long, ctypes.c_double, c_dp, c_lp]
        L.orc_divide_region.restype = ctypes.c_long
        L.orc_divide_region.argtypes = [c_dp, ctypes.c_long, ctypes.c_double, c_dp, ctypes.c_long]
        L.orc_nms.restype = ctypes.c_long
        L.orc_nms.argtypes = [c_fp, ctypes.c_long, c_lp, ctypes.c_double, c_lp]
        L.orc_bbox_overlaps.restype = None
        L.orc_bbox_overlaps.argtypes = [c_dp, ctypes.c_long, c_dp, ctypes.c_long, c_dp]
        L.orc_roi_pool.restype = None
        L.orc_roi_pool.argtypes = [c_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_fp,
                                   ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_float, c_fp]
        L.orc_fc.restype = None
        L.orc_fc.argtypes = [c_fp, ctypes.c_long, ctypes.c_long, c_fp, c_fp, ctypes.c_long,
                             ctypes.c_int, c_fp]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _lp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_long))


# --------------------------------------------------------------------------
# Config: the keys the path reads (lib/detect/config.py:113-216).
# --------------------------------------------------------------------------
class OracleCfg(object):
    DEDUP_BOXES = 1. / 16.          # config.py:206
    EPS = 1e-14                     # config.py:216
    MIN_SIDE = 10                   # SEAR.MIN_SIDE config.py:186
    BATCH_SIZE = 10000              # SEAR.BATCH_SIZE config.py:189
    NUM_PROPOSALS = 300             # TEST.NUM_PROPOSALS config.py:133
    FIXED_PROPOSAL_NUM = True       # SEAR.FIXED_PROPOSAL_NUM config.py:172
    Tc = 0.05                       # SEAR.Tc config.py:171
    Tz = 0.0                        # set by cfg_set_mode config.py:272-280
    NUM_SUBREG = 11                 # len(SEAR.SUBREGION) config.py:149-155
    POOLED = 7                      # test_fc.prototxt:20-21
    SPATIAL_SCALE = 0.0625          # test_fc.prototxt:22

    def __init__(self, **kw):
        for k, v in kw.items():
            if not hasattr(OracleCfg, k):
                raise KeyError(k)
            setattr(self, k, v)


# --------------------------------------------------------------------------
# Native pieces (Cython in the reference).
# --------------------------------------------------------------------------
def divide_children(regions):
    """Children of every parent, before dedup (lib/utils/div.pyx:31-74)."""
    regions = np.ascontiguousarray(regions, dtype=np.float64)
    P = regions.shape[0]
    if P == 0:
        return np.zeros((0, 4))
    lens = np.maximum(regions[:, 2] - regions[:, 0], regions[:, 3] - regions[:, 1]) + 1.0
    shorts = np.minimum(regions[:, 2] - regions[:, 0], regions[:, 3] - regions[:, 1]) + 1.0
    cap = int(np.sum(3 * np.floor(lens / (shorts / 2)) + 2)) + 16
    out = np.zeros((cap, 4))
    n = lib().orc_divide_children(_dp(regions), P, _dp(out), cap)
    assert n >= 0
    return out[:n].copy()


def sift_dup(regions, min_height, return_index=False):
    """lib/utils/div.pyx:78-89."""
    regions = np.ascontiguousarray(regions, dtype=np.float64)
    C = regions.shape[0]
    out = np.zeros((max(C, 1), 4))
    idx = np.zeros(max(C, 1), dtype=np.int64)
    n = lib().orc_sift_dup(_dp(regions), C, float(min_height), _dp(out), _lp(idx))
    if return_index:
        return out[:n].copy(), idx[:n].copy()
    return out[:n].copy()


def divide_region(regions, min_side=10):
    """lib/detect/test.py:153-161 -> lib/utils/div.pyx:15-76."""
    return sift_dup(divide_children(regions), float(min_side))


def sift_dup_numpy(regions, min_height):
    """The same function written with the reference's NumPy calls (div.pyx:85-89);
    cross-checks the int64-key C version."""
    v = np.array([1, 1e3, 1e6, 1e9], dtype=np.float64)
    hashes = np.round(regions / min_height).dot(v)
    _, index = np.unique(hashes, return_index=True)
    return regions[index, :]


def nms(dets, thresh):
    """lib/utils/nms.pyx:17-68.  Returns a list of kept original indices."""
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    N = dets.shape[0]
    if N == 0:
        return []
    order = np.ascontiguousarray(dets[:, 4].argsort()[::-1], dtype=np.int64)   # nms.pyx:25
    keep = np.zeros(N, dtype=np.int64)
    n = lib().orc_nms(_fp(dets), N, _lp(order), float(thresh), _lp(keep))
    return [int(k) for k in keep[:n]]


def bbox_overlaps(boxes, query):
    """lib/utils/bbox.pyx:132-172."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float64)
    query = np.ascontiguousarray(query, dtype=np.float64)
    out = np.zeros((boxes.shape[0], query.shape[0]))
    lib().orc_bbox_overlaps(_dp(boxes), boxes.shape[0], _dp(query), query.shape[0], _dp(out))
    return out


# --------------------------------------------------------------------------
# Head (Caffe-resident in the reference; PARITY UNPINNED).
# --------------------------------------------------------------------------
def roi_pool(feat, rois, pooled=7, spatial_scale=0.0625):
    """ROIPooling, test_fc.prototxt:14-25.  feat [C,H,W] f32, rois [R,5] f32 ->
    [R, C*pooled*pooled] f32 (Caffe's flattening of [R,C,7,7])."""
    feat = np.ascontiguousarray(feat, dtype=np.float32)
    rois = np.ascontiguousarray(rois, dtype=np.float32)
    C, H, W = feat.shape
    R = rois.shape[0]
    out = np.zeros((R, C * pooled * pooled), dtype=np.float32)
    lib().orc_roi_pool(_fp(feat), C, H, W, _fp(rois), R, pooled, pooled,
                       ctypes.c_float(spatial_scale), _fp(out))
    return out


def fc_plain(x, W, b, relu):
    """Scalar k-ascending InnerProduct (cross-check for the BLAS path)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    W = np.ascontiguousarray(W, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    y = np.zeros((x.shape[0], W.shape[0]), dtype=np.float32)
    lib().orc_fc(_fp(x), x.shape[0], x.shape[1], _fp(W), _fp(b), W.shape[0], int(relu), _fp(y))
    return y


_FC_BACKEND = "numpy"


def set_fc_backend(name, threads=None):
    """Which sgemm stands in for Caffe-CPU's cblas_sgemm in fc(): "numpy" (NumPy's BLAS, the default: what the parity tests
    were pinned with) or "torch" (torch CPU addmm, its threads set with torch.set_num_threads -- SURVEY 8(d)'s CPU
    baseline).  Same arithmetic type and layer definition either way; the summation order differs like any two BLAS
    builds do."""
    global _FC_BACKEND
    if name not in ("numpy", "torch"):
        raise ValueError("fc backend: 'numpy' or 'torch'")
    if name == "torch":
        import torch
        if threads:
            torch.set_num_threads(int(threads))
    _FC_BACKEND = name


def fc(x, W, b, relu):
    """InnerProduct via BLAS sgemm (stand-in for Caffe-CPU's cblas_sgemm)."""
    if _FC_BACKEND == "torch":
        import torch
        xt = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        y = torch.addmm(torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32)), xt,
                        torch.from_numpy(np.ascontiguousarray(W, dtype=np.float32)).t())
        if relu:
            y.clamp_(min=0)
        return y.numpy()
    y = x.astype(np.float32, copy=False) @ W.T
    y += b
    if relu:
        np.maximum(y, 0, out=y)
    return y


def sigmoid(x):
    """Caffe Sigmoid: `1. / (1. + exp(-x))` -- f32 exp, double divide, f32 store."""
    e = np.exp(-x.astype(np.float32, copy=False))
    return (1.0 / (1.0 + e.astype(np.float64))).astype(np.float32)


HEAD_KEYS = ("W6", "b6", "W71", "b71", "W72", "b72", "Was", "bas", "Wab", "bab", "Wz", "bz")


def head_forward(head, feat, rois, pooled=7, spatial_scale=0.0625):
    """test_fc.prototxt:14-232: roi_pool5 -> int6+relu -> {int7_1+relu -> adj_score
    (sigmoid), adj_bbox ; int7_2+relu -> zoom_score (sigmoid)}.  Dropout is the
    identity at TEST.  Returns zoom_prob [R,1], adj_prob [R,11], adj_bbox [R,44]."""
    pool5 = roi_pool(feat, rois, pooled, spatial_scale)
    h6 = fc(pool5, head["W6"], head["b6"], True)
    h71 = fc(h6, head["W71"], head["b71"], True)
    h72 = fc(h6, head["W72"], head["b72"], True)
    adj_prob = sigmoid(fc(h71, head["Was"], head["bas"], False))
    adj_bbox = fc(h71, head["Wab"], head["bab"], False)
    zoom_prob = sigmoid(fc(h72, head["Wz"], head["bz"], False))
    return zoom_prob, adj_prob, adj_bbox


class OracleNet(object):
    """Duck-typed pycaffe-like net (`.blobs[name].reshape`, `.forward(blobs=, **inputs)`),
    the shape `lib/detect/test.py:221-236` drives.  `feat_fn(data_blob)` supplies
    conv5_3 for the 'full' net."""

    class _Blob(object):
        def reshape(self, *shape):
            self.shape = shape

    def __init__(self, head, feat_fn=None, name="oracle_net"):
        self.head = head
        self.feat_fn = feat_fn
        self.name = name
        self.blobs = {k: OracleNet._Blob() for k in ("data", "rois", "conv5_3")}
        self.calls = []

    def forward(self, blobs=None, **kw):
        rois = kw["rois"]
        if "conv5_3" in kw:
            conv = kw["conv5_3"]
        else:
            conv = self.feat_fn(kw["data"])
        z, p, d = head_forward(self.head, conv[0], rois)
        self.calls.append(rois.shape[0])
        out = {"zoom_prob": z, "adj_prob": p, "adj_bbox": d}
        if blobs:
            for b in blobs:
                out[b] = conv
        return out


# --------------------------------------------------------------------------
# lib/detect/test.py restated.
# --------------------------------------------------------------------------
def get_rois_blob(im_rois, scale):
    """test.py:61-97 for a single scale (levels == 0)."""
    im_rois = im_rois.astype(np.float64, copy=False)
    levels = np.zeros((im_rois.shape[0], 1), dtype=np.int64)
    scales = np.array([scale])
    rois = im_rois * scales[levels]                       # test.py:95
    return np.hstack((levels, rois)).astype(np.float32, copy=False)   # test.py:70-71


def roi_dedup(rois_blob, dedup=1. / 16.):
    """test.py:210-218: feature-space dedup.  Returns (index, inv_index).
    cfg.DEDUP_BOXES <= 0 skips the dedup (test.py:211 `if cfg.DEDUP_BOXES > 0:`): identity.
    (The reference's _az_forward then fails at test.py:253, np.hstack of a 1-D and an [R,1] array,
    because z_tb is only raveled inside the dedup branch; _frcnn_forward, test.py:281-313, works.
    The restatement ravels in both cases -- the evident intent.)"""
    if not dedup > 0:
        ident = np.arange(rois_blob.shape[0], dtype=np.int64)
        return ident, ident.copy()
    v = np.array([1, 1e3, 1e6, 1e9, 1e12])
    hashes = np.round(rois_blob * dedup).dot(v)
    _, index, inv_index = np.unique(hashes, return_index=True, return_inverse=True)
    return index, inv_index.ravel()


def bbox_pred(boxes, box_deltas, eps=1e-14):
    """test.py:106-139 (north_star's bbox_transform_inv)."""
    if boxes.shape[0] == 0:
        return np.zeros((0, box_deltas.shape[1]))
    boxes = boxes.astype(np.float64, copy=False)
    widths = boxes[:, 2] - boxes[:, 0] + eps
    heights = boxes[:, 3] - boxes[:, 1] + eps
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    dx = box_deltas[:, 0::4]
    dy = box_deltas[:, 1::4]
    dw = box_deltas[:, 2::4]
    dh = box_deltas[:, 3::4]
    pred_ctr_x = dx * widths[:, np.newaxis] + ctr_x[:, np.newaxis]
    pred_ctr_y = dy * heights[:, np.newaxis] + ctr_y[:, np.newaxis]
    pred_w = np.exp(dw) * widths[:, np.newaxis]          # exp in the deltas' dtype (f32)
    pred_h = np.exp(dh) * heights[:, np.newaxis]
    pred_boxes = np.zeros(box_deltas.shape)
    pred_boxes[:, 0::4] = pred_ctr_x - 0.5 * pred_w
    pred_boxes[:, 1::4] = pred_ctr_y - 0.5 * pred_h
    pred_boxes[:, 2::4] = pred_ctr_x + 0.5 * pred_w
    pred_boxes[:, 3::4] = pred_ctr_y + 0.5 * pred_h
    return pred_boxes


def clip_boxes(boxes, im_shape):
    """test.py:141-151 (in place)."""
    boxes[:, 0::4] = np.maximum(boxes[:, 0::4], 0)
    boxes[:, 1::4] = np.maximum(boxes[:, 1::4], 0)
    boxes[:, 2::4] = np.minimum(boxes[:, 2::4], im_shape[1] - 1)
    boxes[:, 3::4] = np.minimum(boxes[:, 3::4], im_shape[0] - 1)
    return boxes


def unwrap_adj_pred(boxes, scores, min_side=10):
    """test.py:171-187."""
    scores = scores.ravel()
    x1 = boxes[:, 0::4].ravel()
    y1 = boxes[:, 1::4].ravel()
    x2 = boxes[:, 2::4].ravel()
    y2 = boxes[:, 3::4].ravel()
    boxes = np.vstack((x1, y1, x2, y2)).transpose()
    heights = boxes[:, 3] - boxes[:, 1] + 1
    widths = boxes[:, 2] - boxes[:, 0] + 1
    sides = np.minimum(heights, widths)
    keep_inds = np.where(sides >= min_side)[0]
    return boxes[keep_inds, :], scores[keep_inds]


def az_forward(net, im_shape, scale, all_boxes, conv, cfg, data_blob=None, trace=None):
    """test.py:189-257.  `im_shape` = original (H, W); `scale` = im_scale of the single
    test scale; `data_blob` = what _get_image_blob would have produced (only consumed
    by net['full'])."""
    batch = cfg.BATCH_SIZE
    num_batches = int(np.ceil(all_boxes.shape[0] / float(batch)))
    zScores = np.zeros((0,))
    aBBox = np.zeros((0, 4))
    cScores = np.zeros((0,))
    for bid in range(num_batches):
        start = batch * bid
        end = min(all_boxes.shape[0], batch * (bid + 1))
        boxes = all_boxes[start:end, 0:4]
        rois = get_rois_blob(boxes, scale)
        index, inv_index = roi_dedup(rois, cfg.DEDUP_BOXES)
        rois = rois[index, :]
        boxes = boxes[index, :]
        if conv is None or "fc" not in net:
            net["full"].blobs["rois"].reshape(*rois.shape)
            out = net["full"].forward(data=data_blob, rois=rois.astype(np.float32, copy=False),
                                      blobs=["conv5_3"])
            conv = {"conv5_3": out["conv5_3"]}
        else:
            net["fc"].blobs["conv5_3"].reshape(*conv["conv5_3"].shape)
            net["fc"].blobs["rois"].reshape(*rois.shape)
            out = net["fc"].forward(rois=rois.astype(np.float32, copy=False), **conv)
        z_tb = out["zoom_prob"]
        pred_scores = out["adj_prob"]
        box_deltas = out["adj_bbox"]
        pred_boxes = bbox_pred(boxes, box_deltas, cfg.EPS)
        pred_boxes = clip_boxes(pred_boxes, im_shape)
        pred_scores = pred_scores[inv_index, :]
        pred_boxes = pred_boxes[inv_index, :]
        z_tb = z_tb[inv_index].ravel()
        a_tb, c_tb = unwrap_adj_pred(pred_boxes, pred_scores, cfg.MIN_SIDE)
        if trace is not None:
            trace.append({"R": end - start, "U": len(index)})
        zScores = np.hstack((zScores, z_tb))
        aBBox = np.vstack((aBBox, a_tb))
        cScores = np.hstack((cScores, c_tb))
    return zScores, aBBox, cScores, conv


def num_levels(height, width, min_side=10):
    """test.py:365-368: K = int(log2(side / MIN_SIDE) + 1) with Python-2 integer
    division of the two integers."""
    side = int(min(height, width))
    return int(np.log2(side // int(min_side)) + 1.0)


def top_k(Y, aScores, num_proposals):
    """test.py:397-401."""
    indA = np.argsort(-aScores)
    max_num = np.minimum(num_proposals, Y.shape[0])
    return Y[indA[:max_num], :], indA[:max_num]


def im_propose(net, im_shape, scale, cfg, data_blob=None, num_proposals=None,
               return_trace=False):
    """test.py:346-414.  `im_shape` = (H, W[, 3]) of the ORIGINAL image."""
    height, width = int(im_shape[0]), int(im_shape[1])
    B = np.array([[0, 0, width - 1.0, height - 1.0]])
    Y = np.zeros((0, 4))
    aScores = np.zeros((0,))
    num_eval = 0
    K = num_levels(height, width, cfg.MIN_SIDE)
    Tz = cfg.Tz
    conv = None
    levels = []
    k = 0
    for k in range(1, K):
        ftrace = []
        zoom, boxes, c, conv = az_forward(net, (height, width), scale, B, conv, cfg,
                                          data_blob, ftrace)
        num_eval += B.shape[0]
        Y = np.vstack((Y, boxes))
        aScores = np.hstack((aScores, c))
        if k == 1:
            zoom[0] = 1.0
        indZ = np.where(zoom >= Tz)[0]
        Z = B[indZ, :]
        levels.append({"B": B.copy(), "zoom": zoom.copy(), "n_cand": boxes.shape[0],
                       "fwd": ftrace, "indZ": indZ.copy()})
        if Z.shape[0] == 0:
            break
        B = divide_region(Z, cfg.MIN_SIDE)
    if (not cfg.FIXED_PROPOSAL_NUM) and (num_proposals is None):
        indA = np.where(aScores >= cfg.Tc)[0]
        Yout = Y[indA, :]
    else:
        if num_proposals is None:
            num_proposals = cfg.NUM_PROPOSALS
        Yout, indA = top_k(Y, aScores, num_proposals)
    if return_trace:
        return Yout, {"levels": levels, "Y_all": Y, "aScores": aScores, "indA": indA,
                      "num_eval": num_eval, "depth": k, "conv": conv}
    return Yout


# --------------------------------------------------------------------------
# Fast R-CNN head on the shared map (BASELINE config 3).
# --------------------------------------------------------------------------
def softmax(x):
    """Caffe Softmax over the channel axis in f32: subtract the max, exp, sum, divide."""
    x = x.astype(np.float32, copy=False)
    e = np.exp(x - x.max(axis=1, keepdims=True))
    return (e / e.sum(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32)


def det_head_forward(head, feat, rois, pooled=7, spatial_scale=0.0625):
    """models/Pascal/VGG16/frcnn/test_fc.prototxt:14-145 (PARITY UNPINNED, see module header):
    roi_pool5 -> fc6+relu -> fc7+relu -> {cls_score -> Softmax = cls_prob, bbox_pred}."""
    pool5 = roi_pool(feat, rois, pooled, spatial_scale)
    h6 = fc(pool5, head["W6"], head["b6"], True)
    h7 = fc(h6, head["W7"], head["b7"], True)
    cls_prob = softmax(fc(h7, head["Wc"], head["bc"], False))
    bbox = fc(h7, head["Wb"], head["bb"], False)
    return cls_prob, bbox


class OracleDetNet(object):
    """pycaffe-shaped Fast R-CNN net for `_frcnn_forward` (test.py:289-307)."""

    def __init__(self, head, name="oracle_det"):
        self.head = head
        self.name = name
        self.blobs = {k: OracleNet._Blob() for k in ("data", "rois", "conv5_3")}

    def forward(self, blobs=None, **kw):
        p, b = det_head_forward(self.head, kw["conv5_3"][0], kw["rois"])
        return {"cls_prob": p, "bbox_pred": b}


def frcnn_forward(net, im_shape, scale, all_boxes, num_classes, conv, cfg):
    """test.py:259-318 with a cached conv (the shared path always has one)."""
    batch = cfg.BATCH_SIZE
    num_batches = int(np.ceil(all_boxes.shape[0] / float(batch)))
    all_pred_boxes = np.zeros((0, 4 * num_classes))
    all_scores = np.zeros((0, num_classes))
    for bid in range(num_batches):
        start = batch * bid
        end = min(all_boxes.shape[0], batch * (bid + 1))
        boxes = all_boxes[start:end, 0:4]
        rois = get_rois_blob(boxes, scale)
        index, inv_index = roi_dedup(rois, cfg.DEDUP_BOXES)
        rois = rois[index, :]
        boxes = boxes[index, :]
        net["fc"].blobs["conv5_3"].reshape(*conv["conv5_3"].shape)
        net["fc"].blobs["rois"].reshape(*rois.shape)
        out = net["fc"].forward(rois=rois.astype(np.float32, copy=False), **conv)
        pred_scores = out["cls_prob"]
        box_deltas = out["bbox_pred"]
        pred_boxes = bbox_pred(boxes, box_deltas, cfg.EPS)
        pred_boxes = clip_boxes(pred_boxes, im_shape)
        pred_scores = pred_scores[inv_index, :]
        pred_boxes = pred_boxes[inv_index, :]
        all_scores = np.vstack((all_scores, pred_scores))
        all_pred_boxes = np.vstack((all_pred_boxes, pred_boxes))
    return all_scores, all_pred_boxes


def im_detect_shared(az_net, frcnn_net, im_shape, scale, num_classes, cfg):
    """test.py:432-445."""
    boxes, tr = im_propose(az_net, im_shape, scale, cfg, return_trace=True)
    scores, pred_boxes = frcnn_forward(frcnn_net, im_shape, scale, boxes, num_classes, tr["conv"], cfg)
    return scores, pred_boxes, boxes


def net_shared_select(per_image, num_classes, max_per_image=100):
    """The per-class bookkeeping of test_net_shared (test.py:670-760) as a pure function of what
    im_detect_shared returned per image: per_image = [(scores [R,K] f64, boxes [R,4K] f64), ...].
    Returns (all_boxes[cls][image] float32 [n,5], thresh [K])."""
    import heapq
    num_images = len(per_image)
    max_per_set = 800 // (num_classes - 1) * num_images        # Python-2 integer division (test.py:676)
    thresh = -np.inf * np.ones(num_classes)
    top_scores = [[] for _ in range(num_classes)]
    all_boxes = [[[] for _ in range(num_images)] for _ in range(num_classes)]
    for i, (scores, boxes) in enumerate(per_image):
        for j in range(1, num_classes):
            inds = np.where(scores[:, j] > thresh[j])[0]                     # test.py:725
            cls_scores = scores[inds, j]
            cls_boxes = boxes[inds, j * 4:(j + 1) * 4]
            top_inds = np.argsort(-cls_scores)[:max_per_image]               # test.py:728
            cls_scores = cls_scores[top_inds]
            cls_boxes = cls_boxes[top_inds, :]
            for val in cls_scores:                                           # test.py:732-733
                heapq.heappush(top_scores[j], val)
            if len(top_scores[j]) > max_per_set:                             # test.py:736-739
                while len(top_scores[j]) > max_per_set:
                    heapq.heappop(top_scores[j])
                thresh[j] = top_scores[j][0]
            all_boxes[j][i] = np.hstack((cls_boxes, cls_scores[:, np.newaxis])).astype(np.float32, copy=False)
    for j in range(1, num_classes):                                          # test.py:755-758
        for i in range(num_images):
            inds = np.where(all_boxes[j][i][:, -1] > thresh[j])[0]
            all_boxes[j][i] = all_boxes[j][i][inds, :]
    return all_boxes, thresh


def apply_nms(all_boxes, thresh):
    """test.py:467-484 (`dets == []` only ever matched the empty-list placeholder)."""
    num_classes = len(all_boxes)
    num_images = len(all_boxes[0])
    nms_boxes = [[[] for _ in range(num_images)] for _ in range(num_classes)]
    for cls_ind in range(num_classes):
        for im_ind in range(num_images):
            dets = all_boxes[cls_ind][im_ind]
            if isinstance(dets, list) and dets == []:
                continue
            keep = nms(dets, thresh)
            if len(keep) == 0:
                continue
            nms_boxes[cls_ind][im_ind] = dets[keep, :].copy()
    return nms_boxes


# --------------------------------------------------------------------------
# Callers either side of the path: image front-end, recall
# evaluation, zoom-threshold tuner.
# --------------------------------------------------------------------------
def image_blob_size(h, w, scale):
    """cv2.resize with fx/fy: dsize = saturate_cast<int>(dim * f), i.e. round half to even."""
    return int(np.rint(h * scale)), int(np.rint(w * scale))


def _linear_taps(n_out, n_in, scale):
    """Source taps of cv2's INTER_LINEAR for float images (OpenCV 2.4/3.x resize(): the
    library is not in /root/reference nor in this image -- published algorithm restated, parity
    unpinned): position (d + 0.5)/scale - 0.5 narrowed to f32; floor and f32 fraction; below
    0 -> tap 0 weight 1; at/after the last pixel -> last pixel weight 1."""
    d = np.arange(n_out, dtype=np.float64)
    f = ((d + 0.5) * (1.0 / scale) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    a = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    s[lo] = 0
    a[lo] = 0
    hi = s >= n_in - 1
    s[hi] = n_in - 1
    a[hi] = 0
    s1 = np.minimum(s + 1, n_in - 1)
    return s, s1, (np.float32(1) - a).astype(np.float32), a


def image_blob(im, pixel_means, scale):
    """_get_image_blob (lib/detect/test.py:27-59) for the single test scale: f32 image minus
    cfg.PIXEL_MEANS, cv2.resize(fx=fy=scale, INTER_LINEAR), HWC -> [1,3,oh,ow]."""
    src = im.astype(np.float32, copy=True)
    src -= np.asarray(pixel_means, dtype=np.float32).reshape(1, 1, 3)
    h, w = src.shape[:2]
    oh, ow = image_blob_size(h, w, scale)
    x0, x1, wx0, wx1 = _linear_taps(ow, w, scale)
    y0, y1, wy0, wy1 = _linear_taps(oh, h, scale)
    hor = (src[:, x0, :] * wx0[None, :, None]).astype(np.float32) + \
          (src[:, x1, :] * wx1[None, :, None]).astype(np.float32)
    hor = hor.astype(np.float32)
    out = (hor[y0] * wy0[:, None, None]).astype(np.float32) + (hor[y1] * wy1[:, None, None]).astype(np.float32)
    return np.ascontiguousarray(out.astype(np.float32).transpose(2, 0, 1)[None])


def recall_gt_overlaps(candidate_boxes, gt_boxes):
    """The matching loop of imdb.evaluate_recall (lib/datasets/imdb.py:123-147): lists of
    per-image candidate / ground-truth boxes -> concatenated per-gt overlaps."""
    gt_overlaps = np.zeros(0)
    for boxes, gts in zip(candidate_boxes, gt_boxes):
        if boxes.shape[0] == 0:
            continue
        overlaps = bbox_overlaps(boxes.astype(np.float64), gts.astype(np.float64))
        per = np.zeros((gts.shape[0]))
        for j in range(gts.shape[0]):
            argmax_overlaps = overlaps.argmax(axis=0)
            max_overlaps = overlaps.max(axis=0)
            gt_ind = max_overlaps.argmax()
            gt_ovr = max_overlaps.max()
            assert gt_ovr >= 0
            box_ind = argmax_overlaps[gt_ind]
            per[j] = overlaps[box_ind, gt_ind]
            overlaps[box_ind, :] = -1
            overlaps[:, gt_ind] = -1
        gt_overlaps = np.hstack((gt_overlaps, per))
    return gt_overlaps


def recall_curve(gt_overlaps):
    """imdb.py:149-158: (ar, sorted gt_overlaps, recalls, thresholds)."""
    num_pos = gt_overlaps.size
    gt_overlaps = np.sort(gt_overlaps)
    step = 0.001
    thresholds = np.minimum(np.arange(0.5, 1.0 + step, step), 1.0)
    recalls = np.zeros_like(thresholds)
    for i, t in enumerate(thresholds):
        recalls[i] = (gt_overlaps >= t).sum() / float(num_pos)
    trapz = getattr(np, "trapezoid", None) or np.trapz
    ar = 2 * trapz(recalls, thresholds)
    return ar, gt_overlaps, recalls, thresholds


def evaluate_recall(candidate_boxes, gt_boxes):
    return recall_curve(recall_gt_overlaps(candidate_boxes, gt_boxes))


def im_propose_tune(net, im_shape, scale, cfg, data_blob=None):
    """The tuner's search (lib/detect/tune.py:256-316): K levels, Tz = 0 for the first
    comparison and cfg.Tz afterwards, no forced root; returns ([Y | score], Bhis)."""
    height, width = int(im_shape[0]), int(im_shape[1])
    B = np.array([[0, 0, width - 1.0, height - 1.0]])
    Bhis = np.zeros((0, 5))
    Y = np.zeros((0, 4))
    aScores = np.zeros((0,))
    K = num_levels(height, width, cfg.MIN_SIDE)
    Tz = 0
    conv = None
    for k in range(K):
        zoom, boxes, c, conv = az_forward(net, (height, width), scale, B, conv, cfg, data_blob, None)
        Y = np.vstack((Y, boxes))
        aScores = np.hstack((aScores, c))
        indZ = np.where(zoom >= Tz)[0]
        Z = B[indZ, :]
        Bhis = np.vstack((Bhis, np.hstack((B, zoom[:, np.newaxis]))))
        if Z.shape[0] == 0:
            break
        B = divide_region(Z, cfg.MIN_SIDE)
        Tz = cfg.Tz
    Yout, indA = top_k(Y, aScores, cfg.NUM_PROPOSALS)
    return np.hstack((Yout, aScores[indA][:, np.newaxis])), Bhis


def tune_thresh(score_lists, max_per_set):
    """tune_thresh's heap (lib/detect/tune.py:326-350) over per-image anchor zoom scores."""
    import heapq
    top_scores = []
    thresh = -np.inf
    for scores in score_lists:
        inds = np.where(scores > thresh)[0]
        for val in scores[inds]:
            heapq.heappush(top_scores, val)
        if len(top_scores) > max_per_set:
            while len(top_scores) > max_per_set:
                heapq.heappop(top_scores)
            thresh = top_scores[0]
    return thresh
