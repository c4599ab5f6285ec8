#!/usr/bin/env python3
"""Use AZ-Net to generate object proposals on an image database -- the MI355X counterpart of
the reference's tools/prop_az.py (same flags, same thresh.pkl input, same proposals.pkl
output).  Differences forced by what exists offline:
  --net   a .caffemodel (read with aznet_hip.caffemodel, no Caffe needed), an .npz with Caffe-layout
          arrays (head: W6,b6,W71,b71,W72,b72,Was,bas,Wab,bab,Wz,bz; optional backbone:
          conv1_1_w/conv1_1_b ...), or `synthetic[:seed]`.
  --def / --def_fc  accepted for command-line compatibility; the layer graph is fixed
          (models/Pascal/VGG16/az-net/test.prototxt, test_fc.prototxt).
  --imdb  `voc_<year>_<split>` (needs data/VOCdevkit<year>), `synthetic_<H>x<W>_<N>` or `npy:<dir>`.
  --recall  (extension) also evaluate recall against the imdb's ground truth
          (imdb.evaluate_recall, lib/datasets/imdb.py:120-159) and store it in proposals.pkl.
With several GPUs: python -m torch.distributed.run --nproc-per-node N tools/prop_az.py ...
shards the images one rank per GPU and gathers the proposals on every rank (RCCL)."""
import _init_paths  # noqa: F401
import argparse
import os
import pickle
import pprint
import sys
import time

import numpy as np

from detect.test import test_proposals, im_propose
from detect.config import cfg, cfg_from_file, cfg_set_mode, cfg_load_thresh, cfg_set_path, get_output_dir


def parse_args():
    parser = argparse.ArgumentParser(description='Use AZ-Net to generate proposals')
    parser.add_argument('--gpu', dest='gpu_id', help='GPU id to use', default=0, type=int)
    parser.add_argument('--def', dest='prototxt', help='(ignored) prototxt of the full net', default=None, type=str)
    parser.add_argument('--def_fc', dest='prototxt_fc', help='(ignored) prototxt of the fc layers', default=None,
                        type=str)
    parser.add_argument('--net', dest='caffemodel', help='AZ-Net weights (.npz) or synthetic[:seed]',
                        default='synthetic', type=str)
    parser.add_argument('--cfg', dest='cfg_file', help='optional config file', default=None, type=str)
    parser.add_argument('--wait', dest='wait', help='wait until net file exists', default=True, type=bool)
    parser.add_argument('--imdb', dest='imdb_name', help='dataset to test', default='synthetic_600x1000_8', type=str)
    parser.add_argument('--thresh', dest='thresh_file', help='file that stores zoom threshold (pickle)', default=None,
                        type=str)
    parser.add_argument('--tz', dest='tz', help='zoom threshold given directly (instead of --thresh)', default=None,
                        type=float)
    parser.add_argument('--exp', dest='exp_dir', help='experiment path', default=None, type=str)
    parser.add_argument('--recall', dest='recall', help='evaluate recall against the ground truth', action='store_true')
    if len(sys.argv) == 1:
        parser.print_help()
        sys.exit(1)
    return parser.parse_args()


def load_net(spec, device):
    from aznet_hip import synth
    from aznet_hip.net import HipAZNet
    from aznet_hip.backbone import VGG16Conv5
    if spec.startswith('synthetic'):
        seed = int(spec.split(':')[1]) if ':' in spec else 1234
        head = synth.make_head(seed=seed, **synth.FULL_DIMS)
        backbone = VGG16Conv5(device='cuda:%d' % device, seed=seed + 1)
        backbone.normalize_output(np.zeros((1, 3, 600, 1000), dtype=np.float32) + 1.0)
        name = 'vgg16_az_net_synthetic_%d' % seed
    elif spec.endswith('.caffemodel'):
        from aznet_hip import caffemodel as cm
        layers = cm.load_caffemodel(spec)
        head = cm.az_head_from_layers(layers)
        backbone = VGG16Conv5(device='cuda:%d' % device, weights=cm.backbone_from_layers(layers))
        name = os.path.splitext(os.path.basename(spec))[0]
    else:
        z = np.load(spec)
        head = {k: z[k] for k in ("W6", "b6", "W71", "b71", "W72", "b72", "Was", "bas", "Wab", "bab", "Wz", "bz")}
        conv = {k[:-2]: (z[k], z[k[:-2] + '_b']) for k in z.files if k.startswith('conv') and k.endswith('_w')}
        backbone = VGG16Conv5(device='cuda:%d' % device, weights=conv or None)
        name = os.path.splitext(os.path.basename(spec))[0]
    return HipAZNet(head, backbone=backbone, device=device, name=name)


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
    if not args.caffemodel.startswith('synthetic'):
        while not os.path.exists(args.caffemodel) and args.wait:
            print('Waiting for {} to exist...'.format(args.caffemodel))
            time.sleep(10)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    device = int(os.environ.get('LOCAL_RANK', args.gpu_id)) if world > 1 else args.gpu_id
    import torch
    torch.cuda.set_device(device)
    net = load_net(args.caffemodel, device)
    nets = {'full': net, 'fc': net}
    from datasets.factory import get_imdb
    imdb = get_imdb(args.imdb_name)
    def report_recall(prop_file):
        with open(prop_file, 'rb') as f:
            prop = pickle.load(f)
        ar, gt_overlaps, recalls, thresholds = imdb.evaluate_recall(prop['boxes'], ctx=net.ctx)
        prop['recall'] = float(recalls[0])
        with open(prop_file, 'wb') as f:
            pickle.dump(prop, f, pickle.HIGHEST_PROTOCOL)
        print('recall@0.5 = {:.4f}, recall@0.7 = {:.4f}, AR = {:.4f} over {:d} gt boxes'.format(
            recalls[0], recalls[200], ar, gt_overlaps.size))

    if world == 1:
        prop_file = test_proposals(nets, imdb)
        if args.recall:
            report_recall(prop_file)
    else:
        import torch.distributed as dist
        from aznet_hip import dist as azdist
        from utils.timer import Timer
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=torch.device('cuda', device))
        n = len(imdb.image_index)
        per = (n + world - 1) // world
        t = Timer()
        local = []
        for j in range(per):                      # rank r owns images r, r + world, ...
            i = min(rank + j * world, n - 1)      # the tail re-runs the last image to keep ranks in step
            im = imdb.image_at(i)
            t.tic()
            Y = im_propose(nets, im)
            t.toc()
            local.append((Y, np.zeros(Y.shape[0], dtype=np.float32)))
        cap = int(cfg.SEAR.NUM_PROPOSALS)
        allp = azdist.gather_proposals(local, cap, device=torch.device('cuda', device))[:n]
        if rank == 0:
            out_dir = get_output_dir(imdb, net)
            os.makedirs(out_dir, exist_ok=True)
            prop = {'boxes': [b for b, _ in allp], 'time': t.average_time, 'recall': 0}
            with open(os.path.join(out_dir, 'proposals.pkl'), 'wb') as f:
                pickle.dump(prop, f, pickle.HIGHEST_PROTOCOL)
            print('wrote', os.path.join(out_dir, 'proposals.pkl'))
            if args.recall:
                report_recall(os.path.join(out_dir, 'proposals.pkl'))
        dist.barrier()
        dist.destroy_process_group()
