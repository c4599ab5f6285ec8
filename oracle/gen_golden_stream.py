#!/usr/bin/env python3
"""Golden vectors for a STREAM of different images at ONE tuned threshold (g14), produced by the REFERENCE's own code
imported from /root/reference in a temp dir (same recipe as gen_golden.py / gen_golden_next.py; nothing of the reference is
copied into the repo):

  1. detect.tune.tune_thresh (lib/detect/tune.py:318-366) over 12 planted-object images (the maps and head of
     synth.make_object_map / make_object_head, SMALL dims; cfg.TRAIN.ANCHORS_PER_IMG = 20) -> the pickled threshold;
  2. detect.test.im_propose (lib/detect/test.py:346-414) on every image at that threshold (moved into the nearest gap
     between two zoom scores of the set: the tuned value IS one of the scores, and BLAS builds differ there by ulps) ->
     Y per image, the rois of every forward.

g14_stream.npz: thresh (the reference's), Tz (used for the searches), per image Y, the unique-roi count of every forward.

Run:  python oracle/gen_golden_stream.py      (this container only; needs /root/reference)
"""
import os
import pickle
import shutil
import sys
import tempfile

import numpy as np
import numpy.ma            # noqa: F401
import scipy.io             # noqa: F401
import scipy.sparse         # noqa: F401
import PIL                  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "az-net_amd", "lib"))
from oracle import gen_golden as gg          # noqa: E402
from oracle import gen_golden_next as gn     # noqa: E402
from aznet_hip import synth                  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
N_IMG, H, W = 12, 600, 1000


def main():
    tmp = tempfile.mkdtemp(prefix="azref_")
    try:
        cdiv, cnms, cbbox, T, C = gg.build_reference(tmp)
        I, U = gn.import_more(tmp)
        head = synth.make_object_head(seed=77, **synth.SMALL_DIMS)
        maps = [synth.make_object_map(j, synth.SMALL_DIMS["C"], 38, 63) for j in range(N_IMG)]
        images = {"img%02d" % i: synth.make_image(100 + i, H, W) for i in range(N_IMG)}
        order = sorted(images.keys())
        cur = [0]
        import cv2
        cv2.imread = lambda path: images[path]

        class StubImdb(object):
            name = "golden_stream"
            image_index = order
            roidb = None

            def image_path_at(self, i):
                return self.image_index[i]

            def gt_roidb(self):
                return None

        full = gg.RecordingNet(head, feat_fn=lambda data: maps[cur[0]], name="golden_net")
        fcn = gg.RecordingNet(head, name="golden_net")
        nets = {"full": full, "fc": fcn}
        # ---- the reference's tuner over the set --------------------------------------------------------------------
        C.cfg_set_mode("Train")
        C.cfg.TEST.MAX_SIZE = 1000
        C.cfg.SEAR.BATCH_SIZE = 10000
        C.cfg_set_path(None)
        C.cfg.TRAIN.ANCHORS_PER_IMG = 20
        ref_tune_propose = U.im_propose
        allz = []

        def spy(net, im):
            cur[0] = next(i for i, k in enumerate(order) if images[k] is im)
            Y5, Bhis = ref_tune_propose(net, im)
            allz.append(Bhis[:, 4].copy())
            return Y5, Bhis

        U.im_propose = spy
        U.tune_thresh(nets, StubImdb())
        out = os.path.join(C.get_output_dir(StubImdb(), full), "thresh.pkl")
        assert out.startswith(tmp)
        thresh = float(pickle.load(open(out, "rb")))
        z = np.sort(np.concatenate(allz))
        i0 = int(np.searchsorted(z, thresh))
        Tz = thresh
        for d in range(len(z)):
            hit = [j for j in (i0 - d, i0 + d) if 1 <= j < len(z) and z[j] - z[j - 1] >= 2e-5]
            if hit:
                Tz = float(0.5 * (z[hit[0]] + z[hit[0] - 1]))
                break
        print("tune_thresh -> %.9f over %d anchors; searches at Tz = %.9f" % (thresh, z.size, Tz))
        # ---- the reference's search on every image at that threshold ------------------------------------------------------
        C.cfg_set_mode("Test", Tz)
        g = {"n_img": np.array(N_IMG), "thresh": np.array(thresh), "Tz": np.array(Tz), "H": np.array(H), "W": np.array(W),
             "anchors_per_img": np.array(20), "pool_size": np.array(z.size)}
        for i, k in enumerate(order):
            cur[0] = i
            full.rec, fcn.rec = [], []
            Y = T.im_propose(nets, images[k])
            rec = full.rec + fcn.rec
            g["Y%d" % i] = Y
            g["calls%d" % i] = np.array([r["rois"].shape[0] for r in rec], dtype=np.int64)
            print("  image", i, "forwards of", [r["rois"].shape[0] for r in rec], "rois; Y", Y.shape)
        np.savez_compressed(os.path.join(GOLD, "g14_stream.npz"), **g)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print("g14 written to", GOLD)


if __name__ == "__main__":
    main()
