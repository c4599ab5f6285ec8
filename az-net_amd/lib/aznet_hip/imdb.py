"""Kept for callers of the first layout: the image databases live in `datasets/` now."""
from datasets.synthetic import SyntheticImdb, NpyDirImdb    # noqa: F401
from datasets.factory import get_imdb                       # noqa: F401
