#!/usr/bin/env python3
"""AZ-Net proposals + Fast R-CNN detection with shared convolutional layers -- the MI355X
counterpart of the reference's tools/test_shared.py (same flags; BASELINE config 3).
As in prop_az.py: --net_az / --net_frcnn take a .caffemodel, an .npz of Caffe-layout arrays or
`synthetic[:seed]`; the --def* prototxt flags are accepted and ignored (the layer graphs are fixed);
--imdb is `voc_<year>_<split>`, `synthetic_<H>x<W>_<N>` or `npy:<dir>`; --tz may replace --thresh."""
import _init_paths  # noqa: F401
import os

import numpy as np

import _cli

FLAGS = [
    ("--def_fc_frcnn", "prototxt_fc_frcnn", "(ignored) prototxt of the Fast R-CNN fc layers", None, str),
    ("--net_frcnn", "caffemodel_frcnn", "Fast R-CNN weights (.caffemodel / .npz) or synthetic[:seed]", "synthetic", str),
    ("--def_az", "prototxt_az", "(ignored) prototxt of the full AZ-Net", None, str),
    ("--def_fc_az", "prototxt_fc_az", "(ignored) prototxt of the AZ-Net fc layers", None, str),
    ("--net_az", "caffemodel_az", "AZ-Net weights (.caffemodel / .npz) or synthetic[:seed]", "synthetic", str),
    ("--imdb", "imdb_name", "dataset to test", "synthetic_600x1000_4", str),
    ("--comp", "comp_mode", "competition mode", None, None),
    # (extension) cfg.TEST.BATCH_IMAGES: the PROPOSALS of consecutive images of one shape in lockstep batches (same detections)
    ("--batch-images", "batch_images", "make the proposals of up to N consecutive images of one shape in lockstep (default 1)", 1, int),
]


def load_det_head(spec):
    """(weights dict for az_load_det_head, net name) from a --net_frcnn value."""
    from aznet_hip import synth
    stem = os.path.splitext(os.path.basename(spec))[0]
    if spec.startswith("synthetic"):
        seed = int(spec.split(":")[1]) if ":" in spec else 4242
        return synth.make_det_head(seed=seed, **synth.FULL_DET_DIMS), "vgg16_frcnn_synthetic_%d" % seed
    if spec.endswith(".caffemodel"):
        from aznet_hip import caffemodel as cm
        return cm.det_head_from_layers(cm.load_caffemodel(spec)), stem
    z = np.load(spec)
    return {k: z[k] for k in ("W6", "b6", "W7", "b7", "Wc", "bc", "Wb", "bb")}, stem


def main():
    args = _cli.parse("Detect objects with AZ-Net proposals and Fast R-CNN on shared conv layers",
                      [_cli.COMMON, _cli.THRESH, FLAGS])
    cfg = _cli.setup_cfg(args, "Test")
    cfg.TEST.BATCH_IMAGES = max(1, int(getattr(args, "batch_images", 1) or 1))
    for f in (args.caffemodel_az, args.caffemodel_frcnn):
        if not f.startswith("synthetic"):
            _cli.wait_for(f, args.wait)
    import torch
    torch.cuda.set_device(args.gpu_id)
    from prop_az import load_net
    from aznet_hip.net import HipDetNet
    from datasets.factory import get_imdb
    from detect.test import test_net_shared
    az_net = load_net(args.caffemodel_az, args.gpu_id, tuned=bool(getattr(args, "tune_backbone", False)))
    det_head, det_name = load_det_head(args.caffemodel_frcnn)
    imdb = get_imdb(args.imdb_name)
    if hasattr(imdb, "competition_mode"):
        imdb.competition_mode(args.comp_mode)
    test_net_shared({"full": az_net, "fc": az_net}, {"fc": HipDetNet(det_head, az_net, name=det_name)}, imdb)


if __name__ == "__main__":
    main()
