"""Image databases for the proposal / detection / tuning drivers (reference: lib/datasets).
Only what those drivers touch is here: names, image paths, ground-truth boxes and the recall
evaluator.  COCO, selective-search roidbs, flipped training copies and the MATLAB evaluation
of the reference's lib/datasets are outside the proposal path."""
import os.path as osp

ROOT_DIR = osp.abspath(osp.join(osp.dirname(__file__), "..", "..", ".."))

from .imdb import imdb                      # noqa: E402,F401
from .pascal_voc import pascal_voc          # noqa: E402,F401
from .synthetic import SyntheticImdb, NpyDirImdb    # noqa: E402,F401
from . import factory                       # noqa: E402,F401
