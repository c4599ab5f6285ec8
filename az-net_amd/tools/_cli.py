"""Command-line plumbing shared by the tools: the reference's flags (tools/prop_az.py:30-54,
tools/set_thresh.py:26-52, tools/test_shared.py) declared as tables, plus the common start-up steps."""
import argparse
import os
import pprint
import sys
import time

# flag, dest, help, default, type   (type None = store_true)
COMMON = [
    ("--gpu", "gpu_id", "GPU id to use", 0, int),
    ("--cfg", "cfg_file", "optional config file", None, str),
    ("--wait", "wait", "wait until the weights file exists", True, bool),
    ("--exp", "exp_dir", "experiment path", None, str),
    # (extension) the backbone in channels_last memory with MIOpen's benchmark search: ~12 % faster convolutions once a shape
    # is tuned, seconds of search the first time a shape is seen -- pays for datasets of few image shapes
    ("--tune-backbone", "tune_backbone", "channels_last VGG16 + MIOpen benchmark search per image shape", None, None),
]
THRESH = [
    ("--thresh", "thresh_file", "file that stores the zoom threshold (pickle)", None, str),
    ("--tz", "tz", "zoom threshold given directly (instead of --thresh)", None, float),
]


def build_parser(description, tables):
    parser = argparse.ArgumentParser(description=description)
    for table in tables:
        for flag, dest, text, default, typ in table:
            if typ is None:
                parser.add_argument(flag, dest=dest, help=text, action="store_true")
            else:
                parser.add_argument(flag, dest=dest, help=text, default=default, type=typ)
    return parser


def parse(description, tables):
    parser = build_parser(description, tables)
    if len(sys.argv) == 1:
        parser.print_help()
        sys.exit(1)
    args = parser.parse_args()
    print("Called with args:")
    print(args)
    return args


def wait_for(path, wait):
    while not os.path.exists(path) and wait:
        print("Waiting for {} to exist...".format(path))
        time.sleep(10)


def setup_cfg(args, mode):
    """cfg_from_file / cfg_set_path / cfg_set_mode in the reference's order; mode 'Test' reads the zoom
    threshold from --tz or the --thresh pickle."""
    from detect.config import cfg, cfg_from_file, cfg_set_mode, cfg_load_thresh, cfg_set_path
    if args.cfg_file is not None:
        cfg_from_file(args.cfg_file)
    cfg_set_path(args.exp_dir)
    if mode == "Test":
        if getattr(args, "tz", None) is not None:
            thresh = args.tz
        else:
            if getattr(args, "thresh_file", None) is None:
                print("error: one of --thresh / --tz is required in Test mode", file=sys.stderr)
                sys.exit(2)
            wait_for(args.thresh_file, args.wait)
            thresh = cfg_load_thresh(args.thresh_file)
        cfg_set_mode("Test", thresh)
    else:
        cfg_set_mode(mode)
    print("Using config:")
    pprint.pprint(cfg)
    return cfg


def ranks():
    """(world, rank, device) from the torch.distributed.run environment, or the single-process default."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    return world, rank
