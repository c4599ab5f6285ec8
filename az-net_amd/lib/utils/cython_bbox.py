"""Drop-in for the reference's `utils.cython_bbox` (built from lib/utils/bbox.pyx) on the GPU:
bbox_overlaps(boxes f64[N,4], query_boxes f64[K,4]) -> f64[N,K] (bbox.pyx:132-172), same typed-
buffer errors.  The training-label helpers of that module (bbox_zoom_labels, bbox_coverage) are
outside the proposal path."""
import numpy as np

from aznet_hip import ffi


def _check(a, name):
    if not isinstance(a, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)" % (name, type(a).__name__))
    if a.dtype != np.float64:
        raise ValueError("Buffer dtype mismatch, expected 'DTYPE_t' but got '%s'" % a.dtype.name)
    if a.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2, got %d)" % a.ndim)


def bbox_overlaps(boxes, query_boxes):
    _check(boxes, "boxes")
    _check(query_boxes, "query_boxes")
    return ffi.default_context().bbox_overlaps(boxes, query_boxes)
