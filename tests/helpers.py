"""Shared test helpers (test infrastructure)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def unpack_list(g, prefix):
    n = g[prefix + "_n"]
    a = g[prefix]
    out, o = [], 0
    for k in n:
        out.append(a[o:o + k])
        o += k
    return out


class _Blob(object):
    def reshape(self, *shape):
        self.shape = shape


class ReplayNet(object):
    """Duck-typed net that replays the head outputs recorded while the REFERENCE's
    im_propose ran (tests/golden/g7_trace_*.npz), asserting it is fed the same rois."""

    def __init__(self, g, which_full, fmap_shape, name="replay"):
        self.g = g
        self.name = name
        self.blobs = {k: _Blob() for k in ("data", "rois", "conv5_3")}
        n = int(g["ncalls"])
        self.calls = [i for i in range(n) if bool(g["c%d_full" % i]) == which_full]
        self.pos = 0
        self.fmap = np.zeros(tuple(fmap_shape), dtype=np.float32)

    def forward(self, blobs=None, **kw):
        i = self.calls[self.pos]
        self.pos += 1
        rois = kw["rois"]
        ref = self.g["c%d_rois" % i]
        assert rois.dtype == np.float32 and rois.shape == ref.shape, (rois.shape, ref.shape)
        assert np.array_equal(rois, ref), "rois fed to the net differ from the reference run"
        out = {k: self.g["c%d_%s" % (i, k)] for k in ("zoom_prob", "adj_prob", "adj_bbox")}
        if blobs:
            for b in blobs:
                out[b] = self.fmap
        return out


def replay_nets(g):
    shp = g["fmap_shape"]
    return {"full": ReplayNet(g, True, shp), "fc": ReplayNet(g, False, shp)}


TRACES = ["a", "b", "c", "d", "e", "f"]


class ReplayDetNet(object):
    """Replays the Fast R-CNN head outputs recorded during the REFERENCE's im_detect_shared run
    (tests/golden/g9_detect_*.npz), asserting it is fed the same rois."""

    def __init__(self, g, name="replay_det"):
        self.g = g
        self.name = name
        self.blobs = {k: _Blob() for k in ("data", "rois", "conv5_3")}
        self.pos = 0

    def forward(self, blobs=None, **kw):
        i = self.pos
        self.pos += 1
        ref = self.g["d%d_rois" % i]
        rois = kw["rois"]
        assert rois.dtype == np.float32 and rois.shape == ref.shape and np.array_equal(rois, ref)
        return {"cls_prob": self.g["d%d_cls_prob" % i], "bbox_pred": self.g["d%d_bbox_pred" % i]}
