"""PASCAL VOC image set (reference: lib/datasets/pascal_voc.py:19-148): the index file, image
paths and XML ground truth of VOCdevkit<year>/VOC<year>.  Results files / MATLAB evaluation are
outside the proposal path."""
import os
import pickle
import xml.etree.ElementTree as ET

import numpy as np

import datasets
from datasets.imdb import imdb

VOC_CLASSES = ("__background__", "aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat",
               "chair", "cow", "diningtable", "dog", "horse", "motorbike", "person", "pottedplant", "sheep",
               "sofa", "train", "tvmonitor")


class pascal_voc(imdb):
    def __init__(self, image_set, year, devkit_path=None):
        imdb.__init__(self, "voc_" + year + "_" + image_set)
        self._year = year
        self._image_set = image_set
        self._devkit_path = devkit_path or os.path.join(datasets.ROOT_DIR, "data", "VOCdevkit" + year)
        self._data_path = os.path.join(self._devkit_path, "VOC" + year)
        self._classes = VOC_CLASSES
        self._class_to_ind = {c: i for i, c in enumerate(VOC_CLASSES)}
        self._image_ext = ".jpg"
        if not os.path.isdir(self._data_path):
            raise IOError("VOC data path does not exist: %s" % self._data_path)
        self._image_index = self._read_index()

    def _read_index(self):
        # <devkit>/VOC<year>/ImageSets/Main/<set>.txt: one image id per line
        path = os.path.join(self._data_path, "ImageSets", "Main", self._image_set + ".txt")
        if not os.path.exists(path):
            raise IOError("image set file does not exist: %s" % path)
        with open(path) as f:
            return [line.strip() for line in f if line.strip()]

    def image_path_from_index(self, index):
        path = os.path.join(self._data_path, "JPEGImages", index + self._image_ext)
        if not os.path.exists(path):
            raise IOError("image does not exist: %s" % path)
        return path

    def image_path_at(self, i):
        return self.image_path_from_index(self._image_index[i])

    @property
    def cache_path(self):
        p = os.path.join(datasets.ROOT_DIR, "data", "cache")
        if not os.path.isdir(p):
            os.makedirs(p)
        return p

    def gt_roidb(self, use_cache=True):
        cache = os.path.join(self.cache_path, self.name + "_gt_roidb.pkl") if use_cache else None
        if cache and os.path.exists(cache):
            with open(cache, "rb") as f:
                return pickle.load(f)
        roidb = [self._load_annotation(ix) for ix in self._image_index]
        if cache:
            with open(cache, "wb") as f:
                pickle.dump(roidb, f, pickle.HIGHEST_PROTOCOL)
        return roidb

    def _load_annotation(self, index):
        """One Annotations/<index>.xml -> 0-based uint16 boxes + class ids (pascal_voc.py:106-143;
        every object is kept, 'difficult' ones included, as in the reference)."""
        tree = ET.parse(os.path.join(self._data_path, "Annotations", index + ".xml"))
        objs = tree.findall("object")
        boxes = np.zeros((len(objs), 4), dtype=np.uint16)
        gt_classes = np.zeros((len(objs),), dtype=np.int32)
        for k, obj in enumerate(objs):
            bb = obj.find("bndbox")
            boxes[k, :] = [float(bb.find(t).text) - 1 for t in ("xmin", "ymin", "xmax", "ymax")]
            gt_classes[k] = self._class_to_ind[obj.find("name").text.lower().strip()]
        return {"boxes": boxes, "gt_classes": gt_classes, "flipped": False}
