#!/usr/bin/env python3
"""AZ-Net proposals + Fast R-CNN detection with shared convolutional layers -- the MI355X
counterpart of the reference's tools/test_shared.py (same flags; BASELINE config 3).
As in prop_az.py: --net_az / --net_frcnn take an .npz of Caffe-layout arrays or `synthetic[:seed]`,
the --def* prototxt flags are accepted and ignored (the layer graphs are fixed), --imdb is
`synthetic_<H>x<W>_<N>` or `npy:<dir>`, and --tz may replace --thresh."""
import _init_paths  # noqa: F401
import argparse
import os
import pprint
import sys
import time

import numpy as np

from detect.test import test_net_shared
from detect.config import cfg, cfg_from_file, cfg_set_mode, cfg_load_thresh, cfg_set_path


def parse_args():
    parser = argparse.ArgumentParser(description='Use Fast-RCNN for object detection')
    parser.add_argument('--gpu', dest='gpu_id', help='GPU id to use', default=0, type=int)
    parser.add_argument('--def_fc_frcnn', dest='prototxt_fc_frcnn', help='(ignored)', default=None, type=str)
    parser.add_argument('--net_frcnn', dest='caffemodel_frcnn', help='Fast R-CNN weights (.npz) or synthetic[:seed]',
                        default='synthetic', type=str)
    parser.add_argument('--def_az', dest='prototxt_az', help='(ignored)', default=None, type=str)
    parser.add_argument('--def_fc_az', dest='prototxt_fc_az', help='(ignored)', default=None, type=str)
    parser.add_argument('--net_az', dest='caffemodel_az', help='AZ-Net weights (.npz) or synthetic[:seed]',
                        default='synthetic', type=str)
    parser.add_argument('--cfg', dest='cfg_file', help='optional config file', default=None, type=str)
    parser.add_argument('--wait', dest='wait', help='wait until net file exists', default=True, type=bool)
    parser.add_argument('--imdb', dest='imdb_name', help='dataset to test', default='synthetic_600x1000_4', type=str)
    parser.add_argument('--comp', dest='comp_mode', help='competition mode', action='store_true')
    parser.add_argument('--thresh', dest='thresh_file', help='file that stores zoom threshold', default=None, type=str)
    parser.add_argument('--tz', dest='tz', help='zoom threshold given directly', default=None, type=float)
    parser.add_argument('--exp', dest='exp_dir', help='experiment path', default=None, type=str)
    if len(sys.argv) == 1:
        parser.print_help()
        sys.exit(1)
    return parser.parse_args()


def load_det_head(spec):
    from aznet_hip import synth
    if spec.startswith('synthetic'):
        seed = int(spec.split(':')[1]) if ':' in spec else 4242
        return synth.make_det_head(seed=seed, **synth.FULL_DET_DIMS), 'vgg16_frcnn_synthetic_%d' % seed
    if spec.endswith('.caffemodel'):
        from aznet_hip import caffemodel as cm
        return cm.det_head_from_layers(cm.load_caffemodel(spec)), os.path.splitext(os.path.basename(spec))[0]
    z = np.load(spec)
    return ({k: z[k] for k in ("W6", "b6", "W7", "b7", "Wc", "bc", "Wb", "bb")},
            os.path.splitext(os.path.basename(spec))[0])


if __name__ == '__main__':
    args = parse_args()
    print('Called with args:')
    print(args)
    if args.cfg_file is not None:
        cfg_from_file(args.cfg_file)
    cfg_set_path(args.exp_dir)
    if args.tz is not None:
        thresh = args.tz
    else:
        while not os.path.exists(args.thresh_file) and args.wait:
            print('Waiting for {} to exist...'.format(args.thresh_file))
            time.sleep(10)
        thresh = cfg_load_thresh(args.thresh_file)
    cfg_set_mode('Test', thresh)
    print('Using config:')
    pprint.pprint(cfg)
    for f in (args.caffemodel_az, args.caffemodel_frcnn):
        while not f.startswith('synthetic') and not os.path.exists(f) and args.wait:
            print('Waiting for {} to exist...'.format(f))
            time.sleep(10)

    import torch
    torch.cuda.set_device(args.gpu_id)
    from prop_az import load_net
    from aznet_hip.net import HipDetNet
    from datasets.factory import get_imdb
    az_net = load_net(args.caffemodel_az, args.gpu_id)
    az_nets = {'full': az_net, 'fc': az_net}
    det_head, det_name = load_det_head(args.caffemodel_frcnn)
    frcnn_nets = {'fc': HipDetNet(det_head, az_net, name=det_name)}
    imdb = get_imdb(args.imdb_name)
    if hasattr(imdb, 'competition_mode'):
        imdb.competition_mode(args.comp_mode)
    test_net_shared(az_nets, frcnn_nets, imdb)
