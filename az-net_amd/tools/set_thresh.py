#!/usr/bin/env python3
"""Set the threshold for zoom indicators -- the MI355X counterpart of the reference's
tools/set_thresh.py (same flags; writes <output_dir>/thresh.pkl, the file tools/prop_az.py
--thresh reads).  --net / --imdb take the same values as tools/prop_az.py here.
With several GPUs (python -m torch.distributed.run --nproc-per-node N tools/set_thresh.py ...)
each rank tunes every N-th image and rank 0 merges the per-rank top scores."""
import _init_paths  # noqa: F401
import os

import _cli

FLAGS = [
    ("--def", "prototxt", "(ignored) prototxt of the full net", None, str),
    ("--def_fc", "prototxt_fc", "(ignored) prototxt of the fc layers", None, str),
    ("--net", "caffemodel", "AZ-Net weights (.caffemodel / .npz) or synthetic[:seed]", "synthetic", str),
    ("--imdb", "imdb_name", "dataset to tune on", "voc_2007_trainval", str),
]


def main():
    args = _cli.parse("Set the zoom threshold of AZ-Net", [_cli.COMMON, FLAGS])
    _cli.setup_cfg(args, "Train")
    if not args.caffemodel.startswith("synthetic"):
        _cli.wait_for(args.caffemodel, args.wait)
    world, rank = _cli.ranks()
    device = int(os.environ.get("LOCAL_RANK", args.gpu_id)) if world > 1 else args.gpu_id
    import torch
    torch.cuda.set_device(device)
    from prop_az import load_net
    from datasets.factory import get_imdb
    from detect.tune import tune_thresh
    net = load_net(args.caffemodel, device, tuned=bool(getattr(args, "tune_backbone", False)))
    nets = {"full": net, "fc": net}
    imdb = get_imdb(args.imdb_name)
    if world == 1:
        tune_thresh(nets, imdb)
        return
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", device_id=torch.device("cuda", device))
    imdb.shard = list(range(rank, len(imdb.image_index), world))

    def gather(top):
        out = [None] * world
        dist.all_gather_object(out, top)
        return out

    tune_thresh(nets, imdb, gather=gather)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
