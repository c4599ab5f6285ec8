"""Zoom-threshold tuning of the reference's `detect.tune` (lib/detect/tune.py) on the MI355X path.

    im_propose(net, im)       tune.py:256-316   -> ([Y | score] [n,5], Bhis [m,5])
    tune_thresh(net, imdb)    tune.py:318-366   -> writes thresh.pkl, returns the threshold

The tuner's search differs from lib/detect/test.py's: it walks K levels (not K-1), compares the
zoom scores of the first level against 0 and of later levels against cfg.SEAR.Tz (0 in 'Train'
mode, tools/set_thresh.py:70), never forces the root, and remembers every anchor region with its
zoom score (Bhis).  It is one az_propose call with the tuner flag.  tune_thresh then needs the
(num_images * cfg.TRAIN.ANCHORS_PER_IMG)-th largest zoom score of the whole set: the scores never
leave HBM (az_tune_begin / az_tune_kth_largest) instead of feeding a Python heap.
"""
import os
import pickle

import numpy as np

from detect.config import cfg, get_output_dir
from detect.test import _im_scale, _get_image_blob
from utils.timer import Timer
from aznet_hip import ffi


def _tune_params(im_shape, scale):
    return ffi.AzContext.make_params(
        im_shape[0], im_shape[1], scale, cfg.SEAR.Tz, num_proposals=cfg.SEAR.NUM_PROPOSALS, fixed_num=True,
        Tc=cfg.SEAR.Tc, dedup=cfg.DEDUP_BOXES, eps=cfg.EPS, min_side=cfg.SEAR.MIN_SIDE,
        batch_size=cfg.SEAR.BATCH_SIZE, tune=True)


def _search(hnet, im, conv=None):
    scales = _im_scale(im.shape)
    if conv is None:
        blob, _ = _get_image_blob(im, hnet)
        hnet.compute_conv(blob)
    else:
        hnet.set_conv(conv[cfg.SEAR.AZ_CONV[0]])
    return hnet.propose(_tune_params(im.shape, scales[0]), want_scores=True, want_stats=True)


def im_propose(net, im, conv=None):
    """tune.py:256-316.  Returns (hstack(Y, scores) [n,5] float64, Bhis [m,5] float64 =
    anchor regions with their zoom scores, level-major)."""
    hnet = net["full"] if isinstance(net, dict) else net
    Y, S, st = _search(hnet, im, conv)
    regions, zoom = hnet.ctx.last_anchors()
    print('{0} proposals, evaluate {1} regions, reaches depth {2}.'.format(Y.shape[0], st.num_eval, st.depth))
    return np.hstack((Y, S.astype(np.float64)[:, np.newaxis])), np.hstack((regions, zoom.astype(np.float64)[:, np.newaxis]))


def tune_thresh(net, imdb, gather=None):
    """Zoom threshold such that on average cfg.TRAIN.ANCHORS_PER_IMG anchors per image score above
    it (tune.py:318-366).  Writes <output_dir>/thresh.pkl (a pickled float, what
    cfg_load_thresh reads) and returns the value.
    gather (multi-GPU): callable(list of float32 arrays) -> list over ranks; each rank tunes its
    shard of the imdb and rank 0 merges the per-rank top scores."""
    hnet = net["full"] if isinstance(net, dict) else net
    num_images = len(imdb.image_index)
    max_per_set = num_images * cfg.TRAIN.ANCHORS_PER_IMG
    output_dir = get_output_dir(imdb, hnet)
    if not os.path.exists(output_dir):
        os.makedirs(output_dir)
    _t = {'im_prop': Timer()}
    mine = getattr(imdb, "shard", None) or range(num_images)
    ctx = hnet.ctx
    ctx.tune_begin(max(1, len(mine)) * 2 * ctx.max_regions)
    # (images read ahead by a worker thread, as in detect.test.test_proposals: reading / decoding an image takes about as long
    #  as the GPU needs for one)
    from detect.test import _prefetched
    images = _prefetched(imdb, list(mine), depth=int(cfg.TEST.get("PREFETCH", 2)))
    for n, i in enumerate(mine):
        im = next(images)
        _t['im_prop'].tic()
        _search(hnet, im)
        _t['im_prop'].toc()
        print('im_tune: {:d}/{:d} {:.3f}s'.format(n + 1, len(mine), _t['im_prop'].average_time))
    if gather is not None:
        tops = gather(ctx.tune_top(max_per_set))
        ctx.tune_begin(max(1, sum(t.size for t in tops)))
        for t in tops:
            ctx.tune_push(t)
    thresh, _ = ctx.tune_kth_largest(max_per_set)
    ctx.tune_end()
    thresh = -np.inf if thresh == float("-inf") else np.float64(thresh)
    print('the threshold is set to {0}'.format(thresh))
    with open(os.path.join(output_dir, 'thresh.pkl'), 'wb') as f:
        pickle.dump(thresh, f, pickle.HIGHEST_PROTOCOL)
    return thresh
