#!/usr/bin/env python3
"""Golden vectors for the harness row, produced by the REFERENCE's own
`test_proposals` (lib/detect/test.py:486-539) and `test_net_shared` (:670-778) imported from
/root/reference in a temp dir (recipe of gen_golden.py; nothing of the reference is copied into the repo).

A 2-image stub imdb (synthetic 375x500 images behind a stub cv2.imread), the seed-77 small AZ head and the
seed-99 small detection head on the CPU (oracle nets), cfg.TEST.NUM_PROPOSALS = 100.  Recorded:

  g13_harness.npz
    prop_boxes<i>            proposals.pkl['boxes'][i] as the reference pickled them (float64 [n,4])
    prop_keys, prop_recall   keys of the pickled dict, its 'recall'
    prop_relpath             proposals.pkl relative to cfg.ROOT_DIR
    prop_stdout              what test_proposals printed (times replaced by 0.000)
    det_scores<i>, det_boxes<i>   what im_detect_shared returned per image inside test_net_shared
    det_all_<j>_<i>          detections.pkl[j][i] (float32 [n,5]); det_nms_<j>_<i> what evaluate_detections got
    det_relpath, det_stdout

Run:  python oracle/gen_golden_harness.py     (this container only; needs /root/reference)
"""
import contextlib
import io
import os
import pickle
import re
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "az-net_amd", "lib"))
from oracle import gen_golden as gg          # noqa: E402
from oracle import az_oracle as orc          # noqa: E402
from aznet_hip import synth                  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
H, W, N_IMG, N_PROP, TZ = 375, 500, 2, 100, 0.0


def scrub(text):
    """Timings are the only run-dependent part of what the harness prints."""
    return re.sub(r"\d+\.\d{3}s", "0.000s", text)


def main():
    tmp = tempfile.mkdtemp(prefix="azref_")
    try:
        cdiv, cnms, cbbox, T, C = gg.build_reference(tmp)
        import cv2                                            # the stub module of build_reference
        cv2.imread = lambda path: synth.make_image(int(path.rsplit("/", 1)[1]), H, W)
        C.cfg.ROOT_DIR = os.path.join(tmp, "root")
        C.cfg_set_path("harness")
        C.cfg.TEST.NUM_PROPOSALS = N_PROP
        C.cfg_set_mode("Test", TZ)
        scale = 600.0 / min(H, W)
        fh, fw = synth.conv_out_size(int(round(H * scale))), synth.conv_out_size(int(round(W * scale)))
        head = synth.make_head(seed=77, **synth.SMALL_DIMS)
        dhead = synth.make_det_head(seed=99, **synth.SMALL_DET_DIMS)

        def fmap_of(data):                                    # the 'full' net's conv5_3: one seeded map per image
            fmap_of.calls += 1
            return synth.make_feature_map(40 + fmap_of.cur, synth.SMALL_DIMS["C"], fh, fw)
        fmap_of.calls = 0
        fmap_of.cur = 0

        class Imdb(object):
            name = "stub_2img"
            image_index = list(range(N_IMG))
            num_classes = 21
            classes = ["c%d" % i for i in range(21)]

            def image_path_at(self, i):
                fmap_of.cur = i                               # the harness reads image i next
                return "synthetic:/%d" % i

            def evaluate_detections(self, nms_dets, output_dir):
                self.nms_dets = nms_dets
                self.eval_dir = output_dir

        full = orc.OracleNet(head, feat_fn=fmap_of, name="az_small")
        fcn = orc.OracleNet(head, name="az_small_fc")
        nets = {"full": full, "fc": fcn}
        imdb = Imdb()
        g = {"H": np.array(H), "W": np.array(W), "n_img": np.array(N_IMG), "n_prop": np.array(N_PROP),
             "Tz": np.array(TZ), "scale": np.array(scale)}

        # ---- test_proposals -----------------------------------------------------------------
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            T.test_proposals(nets, imdb)
        out_dir = C.get_output_dir(imdb, full)
        pf = os.path.join(out_dir, "proposals.pkl")
        with open(pf, "rb") as f:
            prop = pickle.load(f)
        g["prop_keys"] = np.array(sorted(prop.keys()))
        g["prop_recall"] = np.array(prop["recall"])
        g["prop_relpath"] = np.array(os.path.relpath(pf, C.cfg.ROOT_DIR))
        g["prop_stdout"] = np.array(scrub(buf.getvalue()))
        assert isinstance(prop["time"], float)
        for i in range(N_IMG):
            assert prop["boxes"][i].dtype == np.float64
            g["prop_boxes%d" % i] = prop["boxes"][i]

        # ---- test_net_shared ----------------------------------------------------------------
        det = orc.OracleDetNet(dhead)
        rec = []
        inner = T.im_detect_shared

        def recording(az_net, frcnn_net, im, num_classes):
            s, b = inner(az_net, frcnn_net, im, num_classes)
            rec.append((s.copy(), b.copy()))
            return s, b
        T.im_detect_shared = recording
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            T.test_net_shared(nets, {"fc": det}, imdb)
        T.im_detect_shared = inner
        df = os.path.join(out_dir, "detections.pkl")
        with open(df, "rb") as f:
            all_boxes = pickle.load(f)
        g["det_relpath"] = np.array(os.path.relpath(df, C.cfg.ROOT_DIR))
        g["det_stdout"] = np.array(scrub(buf.getvalue()))
        assert imdb.eval_dir == out_dir and len(all_boxes) == 21 and len(rec) == N_IMG
        for i in range(N_IMG):
            g["det_scores%d" % i], g["det_boxes%d" % i] = rec[i]
            for j in range(1, 21):
                a = all_boxes[j][i]
                assert a.dtype == np.float32 and a.shape[1] == 5
                g["det_all_%d_%d" % (j, i)] = a
                n = imdb.nms_dets[j][i]
                g["det_nms_%d_%d" % (j, i)] = np.zeros((0, 5), np.float32) if isinstance(n, list) else n
        np.savez_compressed(os.path.join(GOLD, "g13_harness.npz"), **g)
        print("harness: proposals", [prop["boxes"][i].shape for i in range(N_IMG)], "detections per class/image",
              [[all_boxes[j][i].shape[0] for i in range(N_IMG)] for j in (1, 2, 20)])
        print(g["prop_stdout"])
        print(g["det_stdout"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
