#!/usr/bin/env python3
"""Builder's probe: detect.test.test_proposals over synthetic_600x1000_64 (images cached after the first pass, as a dataset's
files in the page cache) at a threshold that prunes the trees, one image per search and with cfg.TEST.BATCH_IMAGES 8 / 16:
seconds per image as proposals.pkl reports them.   python tests/dev/cli_batch_probe.py [tz]"""
import contextlib
import io
import os
import pickle
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "az-net_amd", "lib"))
sys.path.insert(0, os.path.join(REPO, "az-net_amd", "tools"))


def main():
    tz = float(sys.argv[1]) if len(sys.argv) > 1 else 0.4353
    import torch
    torch.cuda.set_device(0)
    from prop_az import load_net
    from datasets.factory import get_imdb
    from detect import config as C
    from detect import test as T
    C.cfg_set_mode("Test", tz)
    C.cfg.EXP_DIR = "cli_batch_probe_%d" % os.getpid()
    net = load_net("synthetic", 0, tuned=True)
    net.ctx.set_lanes(2)
    imdb = get_imdb("synthetic_600x1000_64")
    for i in range(len(imdb.image_index)):
        imdb.image_at(i)
    for nb in (1, 8, 16, 1, 8):
        C.cfg.TEST.BATCH_IMAGES = nb
        times = []
        for rep in range(3):
            with contextlib.redirect_stdout(io.StringIO()):
                pf = T.test_proposals({"full": net, "fc": net}, imdb)
            with open(pf, "rb") as f:
                times.append(float(pickle.load(f)["time"]))
        print("BATCH_IMAGES %2d: %s ms per image" % (nb, ["%.3f" % (t * 1e3) for t in times]))
    import shutil
    shutil.rmtree(os.path.join(C.cfg.ROOT_DIR, "output", C.cfg.EXP_DIR), ignore_errors=True)


if __name__ == "__main__":
    main()
