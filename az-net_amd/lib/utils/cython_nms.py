"""Drop-in for the reference's Cython module `utils.cython_nms`
(lib/utils/nms.pyx:17-68), backed by the HIP NMS kernels behind az_nms."""
import numpy as np

from aznet_hip import ffi


def nms(dets, thresh):
    """dets float32 [N,5] (x1,y1,x2,y2,score) -> list of kept original indices in
    descending-score order; suppression when IoU >= thresh."""
    if not isinstance(dets, np.ndarray) or dets.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2)")
    if dets.dtype != np.float32:
        raise ValueError("Buffer dtype mismatch, expected 'float32_t' but got '%s'" % dets.dtype)
    return [int(i) for i in ffi.default_context().nms(dets, float(thresh))]
