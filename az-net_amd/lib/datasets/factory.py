"""Image databases by name (reference: lib/datasets/factory.py): voc_<year>_<split> as in the
reference, plus the offline stand-ins synthetic_<H>x<W>_<N> and npy:<directory>."""
from datasets.pascal_voc import pascal_voc
from datasets.synthetic import SyntheticImdb, NpyDirImdb

_makers = {}
for _year in ("2007", "2012", "07+12"):
    for _split in (("trainval",) if _year == "07+12" else ("train", "val", "trainval", "test")):
        _makers["voc_%s_%s" % (_year, _split)] = (lambda s=_split, y=_year: pascal_voc(s, y))


def get_imdb(name):
    if name in _makers:
        return _makers[name]()
    if name.startswith("synthetic_"):
        hw, n = name[len("synthetic_"):].split("_")
        h, w = hw.split("x")
        return SyntheticImdb(int(h), int(w), int(n), name=name)
    if name.startswith("npy:"):
        return NpyDirImdb(name[4:])
    raise KeyError("Unknown dataset: %s (COCO readers of the reference are outside the proposal path)" % name)


def list_imdbs():
    return sorted(_makers) + ["synthetic_<H>x<W>_<N>", "npy:<dir>"]
