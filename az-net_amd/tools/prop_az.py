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
  --tune-backbone  (extension) channels_last VGG16 + MIOpen's benchmark search: ~12 % faster convolutions for a shape once it
          has been searched (seconds, the first time it is seen): for datasets of few image shapes; what bench.py runs.
With several GPUs: python -m torch.distributed.run --nproc-per-node N tools/prop_az.py ...
shards the images one rank per GPU and gathers the proposals on every rank (RCCL)."""
import _init_paths  # noqa: F401
import os
import pickle

import numpy as np

import _cli

FLAGS = [
    ("--def", "prototxt", "(ignored) prototxt of the full net", None, str),
    ("--def_fc", "prototxt_fc", "(ignored) prototxt of the fc layers", None, str),
    ("--net", "caffemodel", "AZ-Net weights (.caffemodel / .npz) or synthetic[:seed]", "synthetic", str),
    ("--imdb", "imdb_name", "dataset to test", "synthetic_600x1000_8", str),
    ("--recall", "recall", "also evaluate recall against the imdb's ground truth", None, None),
    # (extension) several ranks: "nccl" = RCCL over xGMI, one rank per GPU (the production setting); "gloo" = the exchange
    # over host tensors -- ranks that share a GPU (fewer GPUs than ranks), or a box without a usable RCCL
    ("--dist-backend", "dist_backend", "torch.distributed backend of a multi-rank run: nccl (default) or gloo", "nccl", str),
    # (extension) cfg.TEST.BATCH_IMAGES: consecutive images of one shape searched in lockstep (one process; same boxes)
    ("--batch-images", "batch_images", "search up to N consecutive images of one shape in lockstep (default 1: one by one)", 1, int),
]


def load_net(spec, device, tuned=False):
    from aznet_hip import synth
    from aznet_hip.net import HipAZNet
    from aznet_hip.backbone import VGG16Conv5 as _VGG
    if tuned:
        import torch
        torch.backends.cudnn.benchmark = True          # (before the first forward: MIOpen's search per convolution shape)
    if os.environ.get("AZ_BACKBONE_DETERMINISTIC", "0") not in ("", "0"):
        # (tests that compare two PROCESSES box for box: MIOpen may otherwise pick convolution algorithms whose summation
        #  order differs from run to run, which moves conv5_3 by ulps and a box at the MIN_SIDE limit in or out)
        import torch
        torch.backends.cudnn.deterministic = True
        if os.environ["AZ_BACKBONE_DETERMINISTIC"] == "2":
            # (MIOpen off altogether: PyTorch's own im2col + GEMM convolution -- slow, and the same bits on every run, also when
            #  processes share the GPU)
            torch.backends.cudnn.enabled = False

    def VGG16Conv5(**kw):
        return _VGG(channels_last_compute=bool(tuned), channels_last_out=bool(tuned), **kw)
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


def main():
    args = _cli.parse("Use AZ-Net to generate proposals", [_cli.COMMON, _cli.THRESH, FLAGS])
    cfg = _cli.setup_cfg(args, "Test")
    if not args.caffemodel.startswith("synthetic"):
        _cli.wait_for(args.caffemodel, args.wait)
    world, rank = _cli.ranks()
    import torch
    device = args.gpu_id
    if world > 1:
        # one rank per GPU; with fewer GPUs than ranks (--dist-backend gloo) the ranks take the GPUs in turn
        n_dev = torch.cuda.device_count()              # (counting devices does not initialise the GPU)
        device = int(os.environ.get("LOCAL_RANK", args.gpu_id))
        if args.dist_backend == "gloo" and n_dev > 0:
            device %= n_dev
    torch.cuda.set_device(device)
    from datasets.factory import get_imdb
    from detect.config import get_output_dir
    from detect.test import test_proposals, im_propose, _propose_start, _propose_finish, _prefetched, _prefetch_depth, _can_queue
    cfg.TEST.BATCH_IMAGES = max(1, int(getattr(args, "batch_images", 1) or 1))
    net = load_net(args.caffemodel, device, tuned=bool(getattr(args, "tune_backbone", False)))
    nets = {"full": net, "fc": net}
    imdb = get_imdb(args.imdb_name)

    def report_recall(prop_file):
        with open(prop_file, "rb") as f:
            prop = pickle.load(f)
        ar, gt_overlaps, recalls, thresholds = imdb.evaluate_recall(prop["boxes"], ctx=net.ctx)
        prop["recall"] = float(recalls[0])
        with open(prop_file, "wb") as f:
            pickle.dump(prop, f, pickle.HIGHEST_PROTOCOL)
        print("recall@0.5 = {:.4f}, recall@0.7 = {:.4f}, AR = {:.4f} over {:d} gt boxes".format(
            recalls[0], recalls[200], ar, gt_overlaps.size))

    if world == 1:
        prop_file = test_proposals(nets, imdb)
        if args.recall:
            report_recall(prop_file)
        return
    # one rank per GPU: rank r owns images r, r + world, ...; proposals are gathered on every rank
    import torch.distributed as dist
    from aznet_hip import dist as azdist
    from utils.timer import Timer
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", device)
    if args.dist_backend == "gloo":
        dist.init_process_group("gloo")
    elif args.dist_backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        raise SystemExit("--dist-backend: nccl or gloo")
    n = len(imdb.image_index)
    mine = azdist.shard_indices(n, rank, world)
    rows = (n + world - 1) // world                       # short ranks pad (no image is run twice)
    # fixed proposal count and nothing appended: the result records go device-to-device into the RCCL send
    # buffer; otherwise (cfg.SEAR.FIXED_PROPOSAL_NUM off / APPEND_BOXES) host records of an agreed capacity
    fixed = bool(cfg.SEAR.FIXED_PROPOSAL_NUM) and not cfg.SEAR.APPEND_BOXES
    gat = azdist.DeviceGather(net.ctx, int(cfg.SEAR.NUM_PROPOSALS), rows, dev) if fixed else None
    t = Timer()
    local = []
    images = _prefetched(imdb, mine, depth=_prefetch_depth())
    nb = int(cfg.TEST.get("BATCH_IMAGES", 1))
    if fixed and _can_queue(net) and nb > 1:
        # --batch-images: the rank's consecutive images of one shape in lockstep batches (detect.test.test_proposals does the
        # same in a one-process run); a batch's records go to the send buffer in one strided device-to-device copy
        import itertools
        from detect.test import _batch_backbones, _batch_launch, _batch_finish, _lockstep_ok

        def groups():
            cur = []
            for _ in range(len(mine)):
                im = next(images)
                if cur and (_lockstep_ok(im.shape) != _lockstep_ok(cur[0].shape) or len(cur) == nb):
                    yield cur
                    cur = []
                cur.append(im)
            if cur:
                yield cur
        pend, j0 = None, 0
        t.tic()
        for grp in itertools.chain(groups(), [None]):
            nxt = _batch_backbones(nets, grp, after=(pend["done"] if pend is not None else None)) if grp is not None else None
            if pend is not None:
                for i in range(pend["n"]):
                    Y = _batch_finish(nets, pend, i)
                    t.toc()
                    t.tic()
                    local.append((Y, np.zeros(Y.shape[0], dtype=np.float32)))
            if nxt is not None:
                _batch_launch(nets, nxt)
                gat.stage_batch(j0, nxt["n"])
                j0 += nxt["n"]
            pend = nxt
    elif fixed and _can_queue(net):
        # one image ahead, as detect.test.test_proposals: image j+1's pipeline (and the staging of its record) is enqueued
        # while the GPU works on image j
        pend = None
        t.tic()
        for j in range(len(mine) + 1):
            nxt = None
            if j < len(mine):
                nxt = _propose_start(nets, next(images), after=(pend["done"] if pend is not None else None),
                                     stage=(lambda j=j: gat.stage(j)))
            if pend is not None:
                Y = _propose_finish(nets, pend)
                t.toc()
                t.tic()
                local.append((Y, np.zeros(Y.shape[0], dtype=np.float32)))
            pend = nxt
    else:
        for j, i in enumerate(mine):
            im = next(images)
            t.tic()
            Y = im_propose(nets, im, stage=(lambda j=j: gat.stage(j)) if fixed else None)
            t.toc()
            local.append((Y, np.zeros(Y.shape[0], dtype=np.float32)))
    allp = gat.gather(len(mine)) if fixed else azdist.gather_proposals(local, device=None if args.dist_backend == "gloo" else dev)
    assert len(allp) == n
    if rank == 0:
        out_dir = get_output_dir(imdb, net)
        os.makedirs(out_dir, exist_ok=True)
        prop_file = os.path.join(out_dir, "proposals.pkl")
        with open(prop_file, "wb") as f:
            pickle.dump({"boxes": [b for b, _ in allp], "time": t.average_time, "recall": 0}, f, pickle.HIGHEST_PROTOCOL)
        print("wrote", prop_file)
        if args.recall:
            report_recall(prop_file)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
