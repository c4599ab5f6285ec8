#!/usr/bin/env python3
"""Golden vectors for the rows next to the hot path, produced by the
REFERENCE's own code imported from /root/reference in a temp dir (same recipe as
gen_golden.py; nothing of the reference is copied into the repo):

  g10_recall.npz      datasets.imdb.evaluate_recall (lib/datasets/imdb.py:120-159) on seeded
                      candidate / ground-truth box lists
  g11_tune_<t>.npz    detect.tune.im_propose (lib/detect/tune.py:256-316) with an injected,
                      recorded net: [Y | score], Bhis
  g12_tune_thresh.npz detect.tune.tune_thresh (tune.py:318-366) over a 3-image stub imdb:
                      the pickled threshold

Run:  python oracle/gen_golden_next.py      (this container only; needs /root/reference)
"""
import os
import pickle
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np
import numpy.ma            # noqa: F401  (before build_reference installs the np.bool / np.float aliases)
import scipy.io             # noqa: F401
import scipy.sparse         # noqa: F401
import PIL                  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "az-net_amd", "lib"))
from oracle import gen_golden as gg          # noqa: E402
from oracle import az_oracle as orc          # noqa: E402
from aznet_hip import synth                  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")


def import_more(tmp):
    """2to3 + import datasets/imdb.py and detect/tune.py of the temp copy."""
    lib = os.path.join(tmp, "py", "lib")
    files = [os.path.join(lib, "datasets", "imdb.py"), os.path.join(lib, "detect", "tune.py")]
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n"] + files,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in files:
        src = open(f).read().expandtabs(8)
        open(f, "w").write(src)
    # an empty 'datasets' package: datasets/__init__.py insists on MATLAB and imports every reader
    pkg = types.ModuleType("datasets")
    pkg.__path__ = [os.path.join(lib, "datasets")]
    pkg.ROOT_DIR = os.path.join(tmp, "py")
    sys.modules["datasets"] = pkg
    if not hasattr(np, "trapz"):
        np.trapz = np.trapezoid
    import datasets.imdb as I
    import detect.tune as U
    assert I.__file__.startswith(tmp) and U.__file__.startswith(tmp)
    return I, U


def random_boxes(rng, n, W, H, lo=10, hi=200):
    x1 = rng.uniform(0, W - lo - 1, n)
    y1 = rng.uniform(0, H - lo - 1, n)
    w = rng.uniform(lo, hi, n)
    h = rng.uniform(lo, hi, n)
    return np.stack([x1, y1, np.minimum(x1 + w, W - 1), np.minimum(y1 + h, H - 1)], 1)


def main():
    tmp = tempfile.mkdtemp(prefix="azref_")
    try:
        cdiv, cnms, cbbox, T, C = gg.build_reference(tmp)
        I, U = import_more(tmp)
        rng = np.random.RandomState(777)

        # ---------------- G10 evaluate_recall ---------------------------------------
        n_img = 12
        cand, gts, classes = [], [], []
        for i in range(n_img):
            k = int(rng.randint(1, 7))
            gt = np.floor(random_boxes(rng, k, 500, 375, 20, 250))
            n = int(rng.randint(k, 60)) if i != 3 else 0              # image 3: no candidates (skipped)
            b = random_boxes(rng, n, 500, 375)
            if n:
                # some candidates are jittered copies of gt boxes, some exact copies (ties / IoU 1)
                m = min(n, k)
                b[:m] = gt[:m] + rng.uniform(-12, 12, (m, 4))
                if i % 4 == 0:
                    b[0] = gt[0]
                if i % 5 == 0 and n > 1 and k > 1:
                    b[1] = b[0]                                       # duplicate candidates
            cls = np.ones(k, dtype=np.int32)
            if i == 6:
                cls[-1] = 0                                           # a non-positive gt class is ignored
            cand.append(b)
            gts.append(gt)
            classes.append(cls)
        db = I.imdb("golden")
        db._image_index = list(range(n_img))
        db._roidb = [{"boxes": gts[i], "gt_classes": classes[i]} for i in range(n_img)]
        ar, gt_overlaps, recalls, thresholds = db.evaluate_recall(cand)
        g = {"n_img": np.array(n_img), "ar": np.array(ar), "gt_overlaps": gt_overlaps, "recalls": recalls,
             "thresholds": thresholds}
        for i in range(n_img):
            g["cand%d" % i] = cand[i]
            g["gt%d" % i] = gts[i]
            g["cls%d" % i] = classes[i]
        np.savez_compressed(os.path.join(GOLD, "g10_recall.npz"), **g)
        print("recall: ar=%.6f, %d gt boxes" % (ar, gt_overlaps.size))

        # ---------------- G11 tuner's im_propose -------------------------------------
        head = synth.make_head(seed=77, **synth.SMALL_DIMS)
        C.cfg_set_mode("Train")                                       # tools/set_thresh.py:70: Tz = 0
        C.cfg.TEST.MAX_SIZE = 1000
        C.cfg.SEAR.BATCH_SIZE = 10000
        cases = [("a", 375, 500, None), ("b", 480, 640, None), ("c", 375, 500, "q60")]
        for tag, H, W, tz in cases:
            im = synth.make_image(11, H, W)
            scale = 600.0 / min(H, W)
            if np.round(scale * max(H, W)) > 1000:
                scale = 1000.0 / max(H, W)
            fh = synth.conv_out_size(int(round(H * scale)))
            fw = synth.conv_out_size(int(round(W * scale)))
            fmap = synth.make_feature_map(6, synth.SMALL_DIMS["C"], fh, fw)

            def run():
                full = gg.RecordingNet(head, feat_fn=lambda data: fmap)
                fcn = gg.RecordingNet(head)
                Y5, Bhis = U.im_propose({"full": full, "fc": fcn}, im)
                return Y5, Bhis, full.rec + fcn.rec

            C.cfg.SEAR.Tz = 0.0
            Y5, Bhis, rec = run()
            if tz is not None:                                        # a non-zero Tz exercises `Tz = cfg.SEAR.Tz`
                z = Bhis[:, 4]
                Tz = float(np.quantile(z, int(tz[1:]) / 100.0))
                Tz += 0.0 if np.abs(z - Tz).min() > 1e-4 else 2.5e-4
                C.cfg.SEAR.Tz = Tz
                Y5, Bhis, rec = run()
            g = {"H": np.array(H), "W": np.array(W), "Tz": np.array(float(C.cfg.SEAR.Tz)), "scale": np.array(scale),
                 "num_proposals": np.array(int(C.cfg.SEAR.NUM_PROPOSALS)), "Y5": Y5, "Bhis": Bhis,
                 "fmap_shape": np.array(fmap.shape), "ncalls": np.array(len(rec))}
            for i, r in enumerate(rec):
                for k in ("rois", "zoom_prob", "adj_prob", "adj_bbox"):
                    g["c%d_%s" % (i, k)] = r[k]
                g["c%d_full" % i] = np.array(r["full"])
            np.savez_compressed(os.path.join(GOLD, "g11_tune_%s.npz" % tag), **g)
            print("tune", tag, (H, W), "Tz=%.6f" % C.cfg.SEAR.Tz, "anchors", Bhis.shape[0], "Y5", Y5.shape)
        C.cfg.SEAR.Tz = 0.0

        # ---------------- G12 tune_thresh ---------------------------------------------
        shapes = [(375, 500), (333, 500), (375, 500)]
        images = {"img%d" % i: synth.make_image(20 + i, h, w) for i, (h, w) in enumerate(shapes)}
        fmaps = {}
        for i, (h, w) in enumerate(shapes):
            s = 600.0 / min(h, w)
            if np.round(s * max(h, w)) > 1000:
                s = 1000.0 / max(h, w)
            fmaps[(int(round(h * s)), int(round(w * s)))] = synth.make_feature_map(
                30 + i, synth.SMALL_DIMS["C"], synth.conv_out_size(int(round(h * s))),
                synth.conv_out_size(int(round(w * s))))
        import cv2
        cv2.imread = lambda path: images[path]

        class StubImdb(object):
            name = "golden_tune"
            image_index = list(images.keys())
            roidb = None

            def image_path_at(self, i):
                return self.image_index[i]

            def gt_roidb(self):
                return None

        full = gg.RecordingNet(head, feat_fn=lambda data: fmaps[(data.shape[2], data.shape[3])], name="golden_net")
        fcn = gg.RecordingNet(head, name="golden_net")
        C.cfg_set_path(None)                                          # tools/set_thresh.py:68
        ref_im_propose = U.im_propose
        seen = []

        def spy(net, im):
            Y5, Bhis = ref_im_propose(net, im)
            seen.append(Bhis.copy())
            return Y5, Bhis

        U.im_propose = spy
        for per_img in (20, 400):
            C.cfg.TRAIN.ANCHORS_PER_IMG = per_img
            full.rec, fcn.rec = [], []
            del seen[:]
            U.tune_thresh({"full": full, "fc": fcn}, StubImdb())
            out = os.path.join(C.get_output_dir(StubImdb(), full), "thresh.pkl")
            assert out.startswith(tmp)
            thresh = pickle.load(open(out, "rb"))
            # Bhis per image = what the heap consumed (tune.py:338-344)
            g = {"anchors_per_img": np.array(per_img), "thresh": np.array(thresh, dtype=np.float64),
                 "shapes": np.array(shapes), "seeds": np.array([20, 21, 22]), "fmap_seeds": np.array([30, 31, 32])}
            for i, b in enumerate(seen):
                g["bhis%d" % i] = b
            print("tune_thresh: ANCHORS_PER_IMG=%d -> %r" % (per_img, thresh))
            np.savez_compressed(os.path.join(GOLD, "g12_tune_thresh_%d.npz" % per_img), **g)
        C.cfg.TRAIN.ANCHORS_PER_IMG = 20
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()
