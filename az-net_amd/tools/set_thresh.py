#!/usr/bin/env python3
"""Set the threshold for zoom indicators -- the MI355X counterpart of the reference's
tools/set_thresh.py (same flags; writes <output_dir>/thresh.pkl, the file tools/prop_az.py
--thresh reads).  --net / --imdb take the same values as tools/prop_az.py here.
With several GPUs (python -m torch.distributed.run --nproc-per-node N tools/set_thresh.py ...)
each rank tunes every N-th image and rank 0 merges the per-rank top scores."""
import _init_paths  # noqa: F401
import argparse
import os
import pprint
import sys
import time

from detect.tune import tune_thresh
from detect.config import cfg, cfg_from_file, cfg_set_mode, cfg_set_path
from prop_az import load_net


def parse_args():
    parser = argparse.ArgumentParser(description='Set the zoom threshold of AZ-Net')
    parser.add_argument('--gpu', dest='gpu_id', help='GPU id to use', default=0, type=int)
    parser.add_argument('--def', dest='prototxt', help='(ignored) prototxt of the full net', default=None, type=str)
    parser.add_argument('--def_fc', dest='prototxt_fc', help='(ignored) prototxt of the fc layers', default=None,
                        type=str)
    parser.add_argument('--net', dest='caffemodel', help='AZ-Net weights (.caffemodel / .npz) or synthetic[:seed]',
                        default='synthetic', type=str)
    parser.add_argument('--cfg', dest='cfg_file', help='optional config file', default=None, type=str)
    parser.add_argument('--wait', dest='wait', help='wait until net file exists', default=True, type=bool)
    parser.add_argument('--imdb', dest='imdb_name', help='dataset to tune on', default='voc_2007_trainval', type=str)
    parser.add_argument('--exp', dest='exp_dir', help='experiment path', default=None, type=str)
    if len(sys.argv) == 1:
        parser.print_help()
        sys.exit(1)
    return parser.parse_args()


if __name__ == '__main__':
    args = parse_args()
    print('Called with args:')
    print(args)
    if args.cfg_file is not None:
        cfg_from_file(args.cfg_file)
    cfg_set_path(args.exp_dir)
    cfg_set_mode('Train')
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
    if world == 1:
        tune_thresh(nets, imdb)
    else:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=torch.device('cuda', device))
        imdb.shard = list(range(rank, len(imdb.image_index), world))

        def gather(top):
            out = [None] * world
            dist.all_gather_object(out, top)
            return out

        tune_thresh(nets, imdb, gather=gather)
        dist.barrier()
        dist.destroy_process_group()
