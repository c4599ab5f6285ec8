"""Offline stand-ins for a dataset: seeded synthetic images (with seeded ground-truth boxes, so
the recall evaluator and the tuner have something to chew on) and a directory of .npy images."""
import os

import numpy as np

from aznet_hip import synth
from datasets.imdb import imdb


class SyntheticImdb(imdb):
    """`synthetic_<H>x<W>_<N>`: N seeded uint8 BGR images (seed = image index)."""

    def __init__(self, height=600, width=1000, num_images=8, name=None):
        self.height, self.width = int(height), int(width)
        imdb.__init__(self, name or "synthetic_%dx%d_%d" % (self.height, self.width, int(num_images)))
        self._image_index = list(range(int(num_images)))
        self._classes = ["__background__"] + ["class%d" % i for i in range(1, 21)]   # VOC: 21

    def image_at(self, i):
        # (seeded generation takes ~10-20 ms per 600x1000 image, more than the GPU needs for it: an image is generated
        #  once and kept, as a file would sit in the page cache; at most 256 images = 460 MB)
        c = self.__dict__.setdefault("_im_cache", {})
        k = self.image_index[i]
        im = c.get(k)
        if im is None:
            im = synth.make_image(k, self.height, self.width)
            if len(c) < 256:
                c[k] = im
        return im

    def image_path_at(self, i):
        return "synthetic://%d" % self.image_index[i]

    def gt_roidb(self):
        out = []
        for idx in self.image_index:
            rng = np.random.RandomState(10007 + idx)
            k = int(rng.randint(1, 6))
            w = rng.uniform(0.1, 0.6, k) * self.width
            h = rng.uniform(0.1, 0.6, k) * self.height
            x1 = rng.uniform(0, self.width - 1 - w)
            y1 = rng.uniform(0, self.height - 1 - h)
            boxes = np.floor(np.stack([x1, y1, x1 + w, y1 + h], 1)).astype(np.uint16)
            out.append({"boxes": boxes, "gt_classes": rng.randint(1, 21, k).astype(np.int32), "flipped": False})
        return out


class NpyDirImdb(imdb):
    """A directory of HxWx3 uint8 BGR arrays saved as .npy; optional <id>_gt.npy = [k,5]
    (x1,y1,x2,y2,class) ground truth."""

    def __init__(self, path, name=None):
        imdb.__init__(self, name or os.path.basename(os.path.normpath(path)))
        self.path = path
        self._image_index = sorted(f[:-4] for f in os.listdir(path) if f.endswith(".npy") and not f.endswith("_gt.npy"))
        self._classes = ["__background__"] + ["class%d" % i for i in range(1, 21)]

    def image_path_at(self, i):
        return os.path.join(self.path, self.image_index[i] + ".npy")

    def image_at(self, i):
        return np.load(self.image_path_at(i))

    def gt_roidb(self):
        out = []
        for ix in self.image_index:
            p = os.path.join(self.path, ix + "_gt.npy")
            g = np.load(p) if os.path.exists(p) else np.zeros((0, 5))
            out.append({"boxes": g[:, :4].astype(np.uint16), "gt_classes": g[:, 4].astype(np.int32), "flipped": False})
        return out
