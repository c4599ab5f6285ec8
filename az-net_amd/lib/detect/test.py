"""Proposal API of the reference's `detect.test` (lib/detect/test.py), on the MI355X path.

Same names, arguments and return types as the reference for the proposal path:
    im_propose(net, im, return_conv=False, num_proposals=None)      test.py:346-414
    test_proposals(net, imdb)                                       test.py:486-539
    apply_nms(all_boxes, thresh)                                    test.py:467-484
    divide_region(regions)                                          test.py:153-161
`net` is a `HipAZNet` (it also answers net['full'] / net['fc'], so the reference's dict of
two nets keeps working).  The level loop itself -- roi projection and dedup, RoIPool, fc
head, decode, filter, zoom selection, divide_region, top-K -- runs inside one az_propose
call on the GPU; this module only prepares the image blob, reads `cfg`, and formats the
result.  There is no CPU implementation here to fall back to.
"""
import os
import pickle

import numpy as np

from detect.config import cfg, get_output_dir
from utils.timer import Timer
from utils.blob import im_list_to_blob
from utils.cython_nms import nms
import utils.cython_div as div
from aznet_hip import ffi


def _im_scale(im_shape):
    """Scale of the single test scale, capped by MAX_SIZE (test.py:40-50)."""
    im_size_min = np.min(im_shape[0:2])
    im_size_max = np.max(im_shape[0:2])
    scales = []
    for target_size in cfg.TEST.SCALES:
        im_scale = float(target_size) / float(im_size_min)
        if np.round(im_scale * im_size_max) > cfg.TEST.MAX_SIZE:
            im_scale = float(cfg.TEST.MAX_SIZE) / float(im_size_max)
        scales.append(im_scale)
    return scales


def _as_uint8(im):
    """cv2.imread hands the reference uint8 BGR; accept integral-valued arrays of other dtypes."""
    if im.dtype == np.uint8:
        return im
    q = np.rint(im)
    if not (np.array_equal(q, im) and q.min() >= 0 and q.max() <= 255):
        raise TypeError("image must hold uint8 pixel values (as cv2.imread returns)")
    return q.astype(np.uint8)


def _get_image_blob(im, net=None):
    """BGR uint8 image -> ([1,3,H,W] float32 mean-subtracted blob, scale factors)
    (test.py:27-59).  Mean subtraction and the cv2.INTER_LINEAR resize run in one HIP kernel
    (az_image_blob_*); with `net` (a HipAZNet that owns a backbone) the blob stays on the GPU.
    Computed once per image; the reference recomputes it at every level."""
    scales = _im_scale(im.shape)
    if len(scales) != 1:
        raise NotImplementedError("one test scale (cfg.TEST.SCALES), as in every config of the reference")
    src = _as_uint8(im)
    if net is not None:
        blob = net.image_blob(src, cfg.PIXEL_MEANS, scales[0])
    else:
        blob = ffi.default_context().image_blob(src, cfg.PIXEL_MEANS, scales[0])
    return blob, np.array(scales)


def divide_region(regions):
    """test.py:153-161."""
    regions = np.ascontiguousarray(regions, dtype=np.float64)
    return div.divide_region(regions, float(cfg.SEAR.MIN_SIDE))


def _params(im_shape, scale, num_proposals):
    fixed = not ((cfg.SEAR.FIXED_PROPOSAL_NUM is False) and (num_proposals is None))
    if num_proposals is None:
        num_proposals = cfg.SEAR.NUM_PROPOSALS
    return ffi.AzContext.make_params(
        im_shape[0], im_shape[1], scale, cfg.SEAR.Tz, num_proposals=num_proposals, fixed_num=fixed,
        Tc=cfg.SEAR.Tc, dedup=cfg.DEDUP_BOXES, eps=cfg.EPS, min_side=cfg.SEAR.MIN_SIDE,
        batch_size=cfg.SEAR.BATCH_SIZE)


def _append_boxes(boxes):
    """test.py:320-344 (off by default, cfg.SEAR.APPEND_BOXES)."""
    num_boxes = boxes.shape[0]
    num_subregs = cfg.SEAR.APPEND_TEMP.shape[2]
    widths = boxes[:, [2]] - boxes[:, [0]]
    heights = boxes[:, [3]] - boxes[:, [1]]
    L = np.hstack((widths, heights, widths, heights))[:, :, np.newaxis]
    delta = np.hstack((boxes[:, [0]], boxes[:, [1]], boxes[:, [0]], boxes[:, [1]]))[:, :, np.newaxis]
    subs = np.transpose((L * cfg.SEAR.APPEND_TEMP) + delta, [2, 0, 1])
    subs = np.ascontiguousarray(subs.reshape((num_boxes * num_subregs, 4)), dtype=np.float64)
    return div._sift_dup(subs, 1 / cfg.DEDUP_BOXES)


def im_propose(net, im, return_conv=False, num_proposals=None, conv=None, stage=None):
    """Generate object proposals with AZ-Net (test.py:346-414).

    net: HipAZNet (or the reference-style dict {'full': HipAZNet, 'fc': HipAZNet})
    im:  HxWx3 uint8/float image, BGR
    conv (extension): a precomputed conv5_3 {name: array/tensor} to skip the backbone.
    stage (extension, multi-GPU): see HipAZNet.propose.
    Returns Y [n,4] float64 (x1,y1,x2,y2 in original pixels), and the conv dict when
    return_conv is set."""
    hnet = net["full"] if isinstance(net, dict) else net
    scales = _im_scale(im.shape)
    if conv is None:
        blob, _ = _get_image_blob(im, hnet)
        conv_t = hnet.compute_conv(blob)
        conv = {name: conv_t for name in cfg.SEAR.FRCNN_CONV}
    else:
        hnet.set_conv(conv[cfg.SEAR.AZ_CONV[0]])
    params = _params(im.shape, scales[0], num_proposals)
    Y, st = hnet.propose(params, want_stats=True, stage=stage)
    if cfg.SEAR.APPEND_BOXES:
        Y = _append_boxes(Y)
        Y[:, 0::4] = np.maximum(Y[:, 0::4], 0)
        Y[:, 1::4] = np.maximum(Y[:, 1::4], 0)
        Y[:, 2::4] = np.minimum(Y[:, 2::4], im.shape[1] - 1)
        Y[:, 3::4] = np.minimum(Y[:, 3::4], im.shape[0] - 1)
    print('{0} proposals, evaluate {1} regions, reaches depth {2}.'
          .format(Y.shape[0], st.num_eval, st.depth))
    if return_conv:
        return Y, conv
    return Y


# ---- the same search as two halves, so that a loop over images keeps the GPU fed -------------------------------------
# im_propose is synchronous (the reference's contract): image -> blob -> backbone -> search -> boxes, the host waiting at
# every arrow.  A dataset loop does not need the boxes of image i before it may START image i+1: _propose_start enqueues an
# image's whole pipeline -- upload + front-end kernel + backbone on torch's stream, the search behind it on the ctx stream,
# ordered on the device by events -- and returns; _propose_finish collects the boxes.  The GPU runs image after image
# without waiting for Python; results are the same bits (same kernels, same order per image).
def _can_queue(hnet, num_proposals=None):
    fixed = not ((cfg.SEAR.FIXED_PROPOSAL_NUM is False) and (num_proposals is None))
    return fixed and getattr(hnet, "backbone", None) is not None


def _propose_start(net, im, num_proposals=None, after=None, stage=None):
    """First half of im_propose.  after: a torch.cuda.Event the image's front-end + backbone wait for on the device (the
    previous image's search: the two would only compete for the same CUs).  stage: see HipAZNet.propose."""
    import torch
    hnet = net["full"] if isinstance(net, dict) else net
    scale = _im_scale(im.shape)
    if len(scale) != 1:
        raise NotImplementedError("one test scale (cfg.TEST.SCALES), as in every config of the reference")
    params = _params(im.shape, scale[0], num_proposals)
    dev = hnet.backbone.device
    if after is not None:
        torch.cuda.current_stream(dev).wait_event(after)
    blob = hnet.image_blob_enqueue(_as_uint8(im), cfg.PIXEL_MEANS, scale[0])
    conv_t = hnet.backbone(blob)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    hnet.ctx.propose_launch(params, fmap=conv_t, producer_event=ev)
    hnet._conv = conv_t
    if stage is not None:
        stage()
    return {"shape": im.shape, "conv": conv_t, "blob": blob, "done": hnet.ctx.record_event()}


def _propose_finish(net, h, return_conv=False):
    """Second half of im_propose: wait for the search launched by _propose_start, format as im_propose does."""
    hnet = net["full"] if isinstance(net, dict) else net
    Y, st = hnet.ctx.propose_fetch(want_stats=True)
    shape = h["shape"]
    if cfg.SEAR.APPEND_BOXES:
        Y = _append_boxes(Y)
        Y[:, 0::4] = np.maximum(Y[:, 0::4], 0)
        Y[:, 1::4] = np.maximum(Y[:, 1::4], 0)
        Y[:, 2::4] = np.minimum(Y[:, 2::4], shape[1] - 1)
        Y[:, 3::4] = np.minimum(Y[:, 3::4], shape[0] - 1)
    print('{0} proposals, evaluate {1} regions, reaches depth {2}.'
          .format(Y.shape[0], st.num_eval, st.depth))
    if return_conv:
        return Y, {name: h["conv"] for name in cfg.SEAR.FRCNN_CONV}
    return Y


def _batch_backbones(net, ims, after=None):
    """Upload + front-end + backbone of the images of one lockstep batch (cfg.TEST.BATCH_IMAGES), enqueued on torch's stream
    behind `after` (the previous batch's search); the event marks the last map complete."""
    import torch
    hnet = net["full"] if isinstance(net, dict) else net
    dev = hnet.backbone.device
    if after is not None:
        torch.cuda.current_stream(dev).wait_event(after)
    convs, blobs = [], []
    for im in ims:
        scale = _im_scale(im.shape)
        if len(scale) != 1:
            raise NotImplementedError("one test scale (cfg.TEST.SCALES), as in every config of the reference")
        blob = hnet.image_blob_enqueue(_as_uint8(im), cfg.PIXEL_MEANS, scale[0])
        blobs.append(blob)
        conv = hnet.backbone(blob)
        if not conv.is_contiguous(memory_format=torch.channels_last):
            # (the layout the batch reads the maps in; converted here, on torch's stream, so that the hand-over below stays an
            #  event wait on the device)
            conv = conv.contiguous(memory_format=torch.channels_last)
        convs.append(conv)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    # (one AzParams per image: the images of a batch may differ in shape)
    params = [_params(im.shape, _im_scale(im.shape)[0], None) for im in ims]
    return {"shapes": [im.shape for im in ims], "n": len(ims), "ims": ims, "convs": convs, "blobs": blobs, "maps_done": ev,
            "params": params}


def _batch_launch(net, h):
    """The batch's search behind its maps (az_batch_launch: every level's rois of all its images in ONE head pass)."""
    hnet = net["full"] if isinstance(net, dict) else net
    hnet.ctx.batch_launch(h["params"], h["convs"], producer_event=h["maps_done"])
    hnet._conv = h["convs"][-1]
    h["done"] = hnet.ctx.batch_record_event()
    return h


def _batch_finish(net, h, i, quiet=False):
    """Image i of the batch, formatted as im_propose formats it (quiet: the line im_propose prints is returned, not printed)."""
    hnet = net["full"] if isinstance(net, dict) else net
    if "results" not in h:
        h["results"] = hnet.ctx.batch_fetch_all(want_stats=True)      # (the whole batch in one call)
    Y, st = h["results"][i]
    shape = h["shapes"][i]
    if cfg.SEAR.APPEND_BOXES:
        Y = _append_boxes(Y)
        Y[:, 0::4] = np.maximum(Y[:, 0::4], 0)
        Y[:, 1::4] = np.maximum(Y[:, 1::4], 0)
        Y[:, 2::4] = np.minimum(Y[:, 2::4], shape[1] - 1)
        Y[:, 3::4] = np.minimum(Y[:, 3::4], shape[0] - 1)
    line = '{0} proposals, evaluate {1} regions, reaches depth {2}.'.format(Y.shape[0], st.num_eval, st.depth)
    if quiet:
        return Y, line
    print(line)
    return Y


def _num_levels(im_shape):
    """K of im_propose (lib/detect/test.py:365-368; Python-2 integer division when MIN_SIDE is integral): the images of a
    lockstep batch must agree on it."""
    side = min(im_shape[0], im_shape[1])
    ms = cfg.SEAR.MIN_SIDE
    q = side // int(ms) if float(ms) == int(ms) and ms >= 1 else side / float(ms)
    return int(np.log2(q) + 1.0) if q >= 1 else 0


def _lockstep_ok(im_shape):
    """Whether an image's search can share the head passes of a lockstep batch: at least three levels (K - 1 >= 3: an image of
    >= 80 px on its short side at MIN_SIDE 10); smaller ones are batched among themselves and searched one by one."""
    return _num_levels(im_shape) - 1 >= 3


def _batched_proposals(net, images, num_images, nb, launch_ahead=True):
    """(image, proposals, conv maps) for every image of the stream `images`, IN ORDER, the proposals made in lockstep batches
    of up to nb images.  A dataset mixes shapes (VOC: 500x375, 375x500, 500x333, ...): the images of a batch may differ in
    shape and in the number of levels of their trees (az_batch_launch_shapes); only images too small for the lockstep form
    (fewer than three levels) are kept apart -- a batch is the next unprocessed image plus the following images of ITS kind,
    within a window of four batches' worth that is read ahead; results (and the per-image line im_propose prints) are handed
    out in dataset order whatever order the batches ran in.  The next batch's front-ends, backbones and (launch_ahead: a context takes two batches per lane) search
    are enqueued before the current batch's images are handed out; launch_ahead=False: the next search only after the last
    image of the current batch has been taken (a caller that runs other kernels of its own on the context between two images
    -- the detection head -- would find them queued behind that search)."""
    win = max(4 * nb, nb)
    buf, done = {}, {}
    state = {"read": 0, "out": 0}

    def next_group():
        while state["read"] < num_images and len(buf) < win:
            buf[state["read"]] = next(images)
            state["read"] += 1
        if not buf:
            return None
        i0 = min(buf)
        key = _lockstep_ok(buf[i0].shape)
        idx = [i for i in sorted(buf) if _lockstep_ok(buf[i].shape) == key][:nb]
        return idx, [buf.pop(i) for i in idx]

    pend = None
    try:
        while True:
            g = next_group()
            nxt = None
            if g is not None:
                nxt = _batch_backbones(net, g[1], after=(pend["done"] if pend is not None else None))
                nxt["idx"] = g[0]
                if launch_ahead:
                    nxt = _batch_launch(net, nxt)
            if pend is not None:
                for i in range(pend["n"]):
                    Y, line = _batch_finish(net, pend, i, quiet=True)
                    done[pend["idx"][i]] = (pend["ims"][i], Y, {name: pend["convs"][i] for name in cfg.SEAR.FRCNN_CONV}, line)
                while state["out"] in done:
                    im, Y, conv, line = done.pop(state["out"])
                    state["out"] += 1
                    print(line)
                    yield im, Y, conv
            if nxt is None:
                break
            pend = nxt if launch_ahead else _batch_launch(net, nxt)
        assert not done and state["out"] == num_images
    finally:
        # (a loop that ends early -- an error, a caller that stops iterating -- leaves no batch in flight on the context)
        hnet = net["full"] if isinstance(net, dict) else net
        if hnet is not None and hasattr(getattr(hnet, "ctx", None), "batch_drain"):
            hnet.ctx.batch_drain()


def _prefetch_depth():
    """Images read ahead by the worker thread: cfg.TEST.PREFETCH, and with lockstep batches (cfg.TEST.BATCH_IMAGES) at least
    two batches' worth -- the host must be able to collect the next batch while the GPU works on the current one."""
    d = int(cfg.TEST.get("PREFETCH", 2))
    nb = int(cfg.TEST.get("BATCH_IMAGES", 1))
    return max(d, 2 * nb) if (nb > 1 and d > 0) else d


def _prefetched(imdb, indices, depth=2):
    """imdb.image_at(i) for i in indices, in order, read up to `depth` images ahead by a worker thread (decoding a JPEG or
    reading an .npy takes as long as the GPU needs for an image).  depth <= 0: read in the caller's thread."""
    def load(i):
        return imdb.image_at(i) if hasattr(imdb, "image_at") else np.load(imdb.image_path_at(i))
    if depth <= 0 or len(indices) <= 1:
        for i in indices:
            yield load(i)
        return
    import queue
    import threading
    q = queue.Queue(maxsize=depth)
    stop = threading.Event()

    def worker():
        try:
            for i in indices:
                if stop.is_set():
                    return
                q.put(("im", load(i)))
            q.put(("end", None))
        except BaseException as e:                                  # noqa: BLE001 -- handed to the consumer
            q.put(("err", e))
    th = threading.Thread(target=worker, name="az-image-prefetch", daemon=True)
    th.start()
    try:
        while True:
            kind, val = q.get()
            if kind == "end":
                return
            if kind == "err":
                raise val
            yield val
    finally:
        stop.set()
        while th.is_alive():                                        # (unblock a worker waiting on a full queue)
            try:
                q.get_nowait()
            except queue.Empty:
                th.join(0.01)


def _frcnn_forward(net, im, all_boxes, num_classes, conv=None):
    """Fast R-CNN head over proposals on the cached conv map (test.py:259-318): one az_detect
    call (roi projection + dedup, RoIPool, fc6/fc7, cls_score softmax, bbox_pred decode + clip for
    every class, un-dedup).  Returns scores [R, K] and boxes [R, 4K] as float64, and conv."""
    dnet = net["fc"] if isinstance(net, dict) else net
    if conv is not None:
        c = conv[cfg.SEAR.FRCNN_CONV[0]]
        if c is not dnet.az_net._conv:
            dnet.az_net.set_conv(c)
    scale = _im_scale(im.shape)[0]
    assert num_classes == dnet.num_classes
    scores, boxes = dnet.detect(np.ascontiguousarray(all_boxes[:, 0:4], dtype=np.float64), scale, im.shape,
                                cfg.DEDUP_BOXES, cfg.SEAR.BATCH_SIZE, cfg.EPS)
    return scores.astype(np.float64), boxes, conv


def im_detect(net, im, boxes, num_classes):
    """Object classes for given proposals (test.py:416-430); the conv map of `im` must already
    be in the shared context (im_propose / set_conv)."""
    scores, pred_boxes, _ = _frcnn_forward(net, im, boxes, num_classes)
    return scores, pred_boxes


def im_detect_shared(az_net, frcnn_net, im, num_classes):
    """AZ-Net proposals + Fast R-CNN detection on shared convolutional layers (test.py:432-445)."""
    boxes, conv = im_propose(az_net, im, return_conv=True)
    scores, pred_boxes, _ = _frcnn_forward(frcnn_net, im, boxes, num_classes, conv)
    return scores, pred_boxes


def apply_nms(all_boxes, thresh):
    """Per-class, per-image NMS over detections (test.py:467-484).  The reference calls nms once per
    (class, image); here all of them go to the GPU in one az_nms_batched call."""
    num_classes = len(all_boxes)
    num_images = len(all_boxes[0])
    nms_boxes = [[[] for _ in range(num_images)] for _ in range(num_classes)]
    where, sets = [], []
    for cls_ind in range(num_classes):
        for im_ind in range(num_images):
            dets = all_boxes[cls_ind][im_ind]
            if isinstance(dets, list) and dets == []:
                continue
            where.append((cls_ind, im_ind))
            sets.append(np.ascontiguousarray(dets, dtype=np.float32))
    if sets:
        keeps = ffi.default_context().nms_batched(sets, float(thresh))
        for (cls_ind, im_ind), keep in zip(where, keeps):
            if len(keep) == 0:
                continue
            nms_boxes[cls_ind][im_ind] = all_boxes[cls_ind][im_ind][keep, :].copy()
    return nms_boxes


def test_proposals(net, imdb):
    """Proposals for every image of an imdb, written as proposals.pkl with the
    reference's layout {'boxes': [n_i x 4 float64], 'time': avg seconds, 'recall': 0}
    (test.py:486-539).  imdb needs .image_index, .name and .image_at(i) or
    .image_path_at(i) (a .npy path; cv2.imread is not available offline)."""
    num_images = len(imdb.image_index)
    prop_boxes = [[] for _ in range(num_images)]
    hnet = net["full"] if isinstance(net, dict) else net
    output_dir = get_output_dir(imdb, hnet)
    if not os.path.exists(output_dir):
        os.makedirs(output_dir)
    _t = {'im_prop': Timer()}
    num_boxes = 0.0
    images = _prefetched(imdb, list(range(num_images)), depth=_prefetch_depth())
    nb = int(cfg.TEST.get("BATCH_IMAGES", 1))
    if nb > 1 and _can_queue(hnet) and cfg.SEAR.FIXED_PROPOSAL_NUM:
        # cfg.TEST.BATCH_IMAGES (an extension: the reference has no such key): up to that many CONSECUTIVE images of one
        # shape walk their zoom trees in lockstep (az_batch_launch).  Every image's boxes are what im_propose gives for it
        # alone, the printed lines and their order are the reference's; while one batch is searched the host enqueues the
        # next batch's front-ends and backbones behind it.
        done_i = 0
        _t['im_prop'].tic()
        for _im, Y, _conv in _batched_proposals(net, images, num_images, nb):
            prop_boxes[done_i] = Y
            done_i += 1
            _t['im_prop'].toc()
            print('im_prop: {:d}/{:d} {:.3f}s'.format(done_i, num_images, _t['im_prop'].average_time))
            _t['im_prop'].tic()
    elif _can_queue(hnet):
        # One image ahead: while the GPU works on image i the host reads image i+1 and enqueues its whole pipeline behind
        # it.  Same boxes, same printed lines in the same order; the timer counts from one finished image to the next
        # (what a per-image tic/toc adds up to when nothing overlaps).
        pend = None
        _t['im_prop'].tic()
        for i in range(num_images + 1):
            nxt = None
            if i < num_images:
                nxt = _propose_start(net, next(images), after=(pend["done"] if pend is not None else None))
            if pend is not None:
                prop_boxes[i - 1] = _propose_finish(net, pend)
                _t['im_prop'].toc()
                print('im_prop: {:d}/{:d} {:.3f}s'.format(i, num_images, _t['im_prop'].average_time))
                _t['im_prop'].tic()
            pend = nxt
    else:
        for i in range(num_images):
            im = next(images)
            _t['im_prop'].tic()
            prop_boxes[i] = im_propose(net, im)
            _t['im_prop'].toc()
            # (num_boxes stays 0.0: the reference's per-image bookkeeping is commented out, test.py:515-526,
            #  so its "On average, 0.0 boxes per image are generated" line is reproduced as is)
            print('im_prop: {:d}/{:d} {:.3f}s'.format(i + 1, num_images, _t['im_prop'].average_time))
    recall = 0            # the reference's recall bookkeeping is commented out (test.py:515-531)
    prop = {'boxes': prop_boxes, 'time': _t['im_prop'].average_time, 'recall': recall}
    prop_file = os.path.join(output_dir, 'proposals.pkl')
    with open(prop_file, 'wb') as f:
        pickle.dump(prop, f, pickle.HIGHEST_PROTOCOL)
    print('The recall is {:.3f}'.format(recall))
    print('On average, {0} boxes per image are generated'.format(num_boxes / num_images))
    print('The average proposal generation time is {:.3f}s'.format(_t['im_prop'].average_time))
    return prop_file


def test_net_shared(sc_net, frcnn_net, imdb):
    """Detection over an imdb with shared conv layers (test.py:670-778): per class keep scores
    above an adaptive threshold, at most 100 per image and `800 / (K-1)` per image on average
    over the set (min-heap), write detections.pkl, apply NMS (cfg.TEST.NMS) and hand the result to
    imdb.evaluate_detections when the imdb has one."""
    import heapq
    num_images = len(imdb.image_index)
    num_classes = imdb.num_classes
    max_per_set = 800 // (num_classes - 1) * num_images          # Python-2 integer division (test.py:676)
    max_per_image = 100
    thresh = -np.inf * np.ones(num_classes)
    top_scores = [[] for _ in range(num_classes)]
    all_boxes = [[[] for _ in range(num_images)] for _ in range(num_classes)]
    num_boxes = 0.0
    hnet = sc_net["full"] if isinstance(sc_net, dict) else sc_net
    output_dir = get_output_dir(imdb, hnet)
    if not os.path.exists(output_dir):
        os.makedirs(output_dir)
    _t = {'im_detect': Timer(), 'misc': Timer()}
    # Images are read ahead by a worker thread, and (fixed proposal count, a backbone on the GPU) image i+1's upload +
    # front-end + backbone + search are enqueued as soon as image i's detections are on the host: the GPU works on image
    # i+1 while Python does image i's per-class bookkeeping below.  Same calls per image, same printed lines in the same
    # order; the reference's loop (test.py:690-737) waits for each image before it reads the next.
    images = _prefetched(imdb, list(range(num_images)), depth=_prefetch_depth())
    queued = _can_queue(hnet) and num_images > 0
    # cfg.TEST.BATCH_IMAGES > 1 (an extension): the proposals of consecutive images of one shape in lockstep batches
    # (az_batch_launch), the detection head image by image as before; same detections, same printed lines
    nb = int(cfg.TEST.get("BATCH_IMAGES", 1))
    batched = nb > 1 and queued and bool(cfg.SEAR.FIXED_PROPOSAL_NUM)
    gen = _batched_proposals(sc_net, images, num_images, nb, launch_ahead=False) if batched else None
    pend, im = None, None
    if queued and not batched:
        im = next(images)
        pend = _propose_start(sc_net, im)
    for i in range(num_images):
        _t['im_detect'].tic()
        if batched:
            im, prop, conv = next(gen)
            # (the image's map is complete -- its search has been fetched --: bound to the context without waiting for torch's
            #  stream, on which the NEXT batch's backbones are already queued)
            c0 = conv[cfg.SEAR.FRCNN_CONV[0]]
            hnet.ctx.set_feature_map(c0, producer_done=True)
            hnet._conv = c0
            scores, boxes, _ = _frcnn_forward(frcnn_net, im, prop, num_classes, conv)
        elif queued:
            prop, conv = _propose_finish(sc_net, pend, return_conv=True)
            scores, boxes, _ = _frcnn_forward(frcnn_net, im, prop, num_classes, conv)
            if i + 1 < num_images:
                im = next(images)
                pend = _propose_start(sc_net, im)
        else:
            im = next(images)
            scores, boxes = im_detect_shared(sc_net, frcnn_net, im, num_classes)
        num_boxes += scores.shape[0]
        _t['im_detect'].toc()
        _t['misc'].tic()
        for j in range(1, num_classes):
            inds = np.where((scores[:, j] > thresh[j]))[0]
            cls_scores = scores[inds, j]
            cls_boxes = boxes[inds, j * 4:(j + 1) * 4]
            top_inds = np.argsort(-cls_scores)[:max_per_image]
            cls_scores = cls_scores[top_inds]
            cls_boxes = cls_boxes[top_inds, :]
            for val in cls_scores:
                heapq.heappush(top_scores[j], val)
            if len(top_scores[j]) > max_per_set:
                while len(top_scores[j]) > max_per_set:
                    heapq.heappop(top_scores[j])
                thresh[j] = top_scores[j][0]
            all_boxes[j][i] = np.hstack((cls_boxes, cls_scores[:, np.newaxis])).astype(np.float32, copy=False)
        _t['misc'].toc()
        print('im_detect: {:d}/{:d} {:.3f}s {:.3f}s'.format(i + 1, num_images, _t['im_detect'].average_time,
                                                          _t['misc'].average_time))
    for j in range(1, num_classes):
        for i in range(num_images):
            inds = np.where(all_boxes[j][i][:, -1] > thresh[j])[0]
            all_boxes[j][i] = all_boxes[j][i][inds, :]
    det_file = os.path.join(output_dir, 'detections.pkl')
    with open(det_file, 'wb') as f:
        pickle.dump(all_boxes, f, pickle.HIGHEST_PROTOCOL)
    print('Applying NMS to all detections')
    nms_dets = apply_nms(all_boxes, cfg.TEST.NMS)
    if hasattr(imdb, "evaluate_detections"):
        print('Evaluating detections')
        imdb.evaluate_detections(nms_dets, output_dir)
    print('The average detection time is {:.3f}s'.format(_t['im_detect'].average_time))
    print('On average, {0} boxes per image are proposed'.format(num_boxes / num_images))
    return nms_dets
