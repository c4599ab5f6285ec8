"""Base image database (reference: lib/datasets/imdb.py): identity, image list, ground truth,
and `evaluate_recall` -- whose box matching runs on the GPU (az_recall_match)."""
import numpy as np

from aznet_hip import ffi


class imdb(object):
    def __init__(self, name):
        self._name = name
        self._classes = []
        self._image_index = []
        self._roidb = None
        self.config = {}

    # -- identity ------------------------------------------------------------------------
    @property
    def name(self):
        return self._name

    @property
    def classes(self):
        return self._classes

    @property
    def num_classes(self):
        return len(self._classes)

    @property
    def image_index(self):
        return self._image_index

    @property
    def num_images(self):
        return len(self.image_index)

    # -- ground truth ----------------------------------------------------------------------
    @property
    def roidb(self):
        """List of per-image dicts with 'boxes' [k,4] and 'gt_classes' [k] (imdb.py:59-68).
        The reference can fill it from several proposal sources; here it is the ground truth."""
        if self._roidb is None:
            self._roidb = self.gt_roidb()
        return self._roidb

    def gt_roidb(self):
        raise NotImplementedError

    def image_path_at(self, i):
        raise NotImplementedError

    def image_at(self, i):
        """BGR uint8 HxWx3 array of image i (cv2.imread's convention, which is not available
        offline): decoded with PIL and flipped RGB -> BGR."""
        from PIL import Image
        with Image.open(self.image_path_at(i)) as im:
            rgb = np.asarray(im.convert("RGB"))
        return np.ascontiguousarray(rgb[:, :, ::-1])

    # -- recall ------------------------------------------------------------------------------
    def evaluate_recall(self, candidate_boxes, ar_thresh=0.5, ctx=None):
        """imdb.evaluate_recall (lib/datasets/imdb.py:120-159): for each image match candidates to
        ground-truth boxes greedily by IoU, then sweep IoU thresholds 0.5..1.0.
        Returns (ar, gt_overlaps (sorted), recalls, thresholds) like the reference."""
        ctx = ctx or ffi.default_context()
        cands, gts = [], []
        for i in range(self.num_images):
            entry = self.roidb[i]
            pos = np.where(entry["gt_classes"] > 0)[0]
            boxes = candidate_boxes[i]
            if boxes.shape[0] == 0:                  # imdb.py:128-129: such images do not count
                continue
            cands.append(np.asarray(boxes, dtype=np.float64)[:, :4])
            gts.append(np.asarray(entry["boxes"])[pos, :].astype(np.float64))
        try:
            gt_overlaps = ctx.recall_match(cands, gts)
        except ffi.AzError as e:
            if e.code == ffi.AZ_ERR_INVALID:         # the reference's `assert(gt_ovr >= 0)`
                raise AssertionError(str(e))
            raise
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
