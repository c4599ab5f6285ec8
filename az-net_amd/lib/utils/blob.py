"""Image-list -> NCHW float32 blob (reference: lib/utils/blob.py:13-29)."""
import numpy as np


def im_list_to_blob(ims):
    """Zero-pad HxWx3 images to a common size and move channels first."""
    max_shape = np.array([im.shape for im in ims]).max(axis=0)
    blob = np.zeros((len(ims), max_shape[0], max_shape[1], max_shape[2]), dtype=np.float32)
    for i, im in enumerate(ims):
        blob[i, 0:im.shape[0], 0:im.shape[1], :] = im
    return blob.transpose((0, 3, 1, 2))
