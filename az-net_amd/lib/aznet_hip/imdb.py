"""Minimal image databases for the proposal driver.  The reference's lib/datasets (VOC /
COCO readers, MATLAB eval) is out of scope and no dataset exists offline; test_proposals
only needs `.name`, `.image_index` and `.image_at(i)` / `.image_path_at(i)`."""
import os

import numpy as np

from aznet_hip import synth


class SyntheticImdb(object):
    """`synthetic_<H>x<W>_<N>`: N seeded uint8 BGR images (seed = image index)."""

    def __init__(self, height=600, width=1000, num_images=8, name=None):
        self.height, self.width = int(height), int(width)
        self.image_index = list(range(int(num_images)))
        self.num_classes = 21            # VOC: background + 20 (models/Pascal)
        self.name = name or "synthetic_%dx%d_%d" % (self.height, self.width, len(self.image_index))

    def image_at(self, i):
        return synth.make_image(self.image_index[i], self.height, self.width)

    def image_path_at(self, i):
        return "synthetic://%d" % self.image_index[i]


class NpyDirImdb(object):
    """A directory of HxWx3 uint8 BGR arrays saved as .npy (cv2.imread is not available)."""

    def __init__(self, path, name=None):
        self.path = path
        self.image_index = sorted(f[:-4] for f in os.listdir(path) if f.endswith(".npy"))
        self.name = name or os.path.basename(os.path.normpath(path))

    def image_path_at(self, i):
        return os.path.join(self.path, self.image_index[i] + ".npy")

    def image_at(self, i):
        return np.load(self.image_path_at(i))


def get_imdb(name):
    """`synthetic_600x1000_8`, or `npy:<directory>`."""
    if name.startswith("synthetic_"):
        hw, n = name[len("synthetic_"):].split("_")
        h, w = hw.split("x")
        return SyntheticImdb(int(h), int(w), int(n), name=name)
    if name.startswith("npy:"):
        return NpyDirImdb(name[4:])
    raise KeyError("unknown imdb %r (datasets of the reference are out of scope: use "
                   "synthetic_<H>x<W>_<N> or npy:<dir>)" % name)
