"""Drop-in for the reference's Cython module `utils.cython_div`
(lib/utils/div.pyx:15-89), backed by the HIP kernels behind az_divide_region /
az_sift_dup.  Same call signatures, dtype strictness and return types."""
import numpy as np

from aznet_hip import ffi


def _check(regions):
    if not isinstance(regions, np.ndarray) or regions.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2)")
    if regions.dtype != np.float64:
        raise ValueError("Buffer dtype mismatch, expected 'DTYPE_t' but got '%s'" % regions.dtype)


def divide_region(regions, min_height):
    """float64 [P,4] parents -> float64 [C',4] children, deduplicated on the
    min_height grid and sorted by hash (div.pyx:15-76)."""
    _check(regions)
    return ffi.default_context().divide_region(regions, float(min_height))


def _sift_dup(regions, min_height):
    """div.pyx:78-89."""
    _check(regions)
    return ffi.default_context().sift_dup(regions, float(min_height))
