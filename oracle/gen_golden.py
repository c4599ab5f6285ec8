#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE'S OWN CODE.

Runs only in the build container (needs /root/reference).  Nothing from the
reference is copied into this repository: its .pyx / .py files are copied to a
throw-away temp directory, made importable under Python 3 / NumPy 2 with the
token-level edits listed below, executed on seeded inputs, and only the
resulting input/output arrays are written to tests/golden/.

Edits applied to the temp copies (NumPy removed the aliases the 2015 code uses):
  nms.pyx:17     `np.float thresh`  -> `double thresh`   (Cython resolves the original to
                 the builtin Python float, i.e. a double-precision comparison at :65)
  nms.pyx:25,28  `np.int_t` -> `np.intp_t` ;  nms.pyx:29 `dtype=np.int` -> `np.intp`
  div.pyx:12, bbox.pyx:17   `DTYPE = np.float` -> `np.float64`
  bbox.pyx:85    `np.bool` -> `bool`
  lib/detect/{test,config}.py, lib/utils/{blob,timer}.py: `lib2to3`, literal TABs in
  config.py -> spaces, `np.float/np.int/np.bool` aliases installed before import,
  stub modules `caffe`, `cv2` (resize returns zeros of the scaled shape), `easydict`.
  test.py:477    `if dets == []:` -> `if isinstance(dets, list) and dets == []:`  (NumPy of 2015 answered
                 `ndarray == []` with a scalar False -- "elementwise comparison failed" -- so the test only
                 ever fired for the empty-list placeholder; NumPy 2 raises a broadcast error instead)

Usage:  python oracle/gen_golden.py            (writes tests/golden/)
"""
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "az-net_amd", "lib"))

from oracle import az_oracle as orc          # noqa: E402
from aznet_hip import synth                  # noqa: E402


def build_reference(tmp):
    """Returns (cython_div, cython_nms, cython_bbox, detect.test module, detect.config module)."""
    cy = os.path.join(tmp, "cy")
    os.makedirs(os.path.join(cy, "utils"))
    for f in ("div.pyx", "nms.pyx", "bbox.pyx"):
        shutil.copy(os.path.join(REF, "lib", "utils", f), os.path.join(cy, "utils", f))
    open(os.path.join(cy, "utils", "__init__.py"), "w").close()

    def sub(path, pairs):
        s = open(path).read()
        for a, b in pairs:
            assert a in s, (path, a)
            s = s.replace(a, b)
        open(path, "w").write(s)

    sub(os.path.join(cy, "utils", "nms.pyx"),
        [("np.float thresh", "double thresh"), ("np.int_t", "np.intp_t"), ("dtype=np.int)", "dtype=np.intp)")])
    sub(os.path.join(cy, "utils", "div.pyx"), [("DTYPE = np.float\n", "DTYPE = np.float64\n")])
    sub(os.path.join(cy, "utils", "bbox.pyx"),
        [("DTYPE = np.float\n", "DTYPE = np.float64\n"), ("dtype=np.bool", "dtype=bool")])
    with open(os.path.join(cy, "setup.py"), "w") as f:
        f.write(
            "from setuptools import setup, Extension\n"
            "from Cython.Build import cythonize\nimport numpy\n"
            "exts=[Extension('utils.cython_%s'%n,['utils/%s.pyx'%n],include_dirs=[numpy.get_include()],"
            "extra_compile_args=['-O2','-ffp-contract=off','-w']) for n in ('div','nms','bbox')]\n"
            "setup(ext_modules=cythonize(exts,language_level=2))\n")
    subprocess.check_call([sys.executable, "setup.py", "-q", "build_ext", "--inplace"], cwd=cy,
                          stdout=subprocess.DEVNULL)

    py = os.path.join(tmp, "py")
    shutil.copytree(os.path.join(REF, "lib"), os.path.join(py, "lib"))
    subprocess.check_call(["chmod", "-R", "u+w", py])
    for so in os.listdir(os.path.join(cy, "utils")):
        if so.endswith(".so"):
            shutil.copy(os.path.join(cy, "utils", so), os.path.join(py, "lib", "utils", so))
    files = [os.path.join(py, "lib", "detect", "test.py"), os.path.join(py, "lib", "detect", "config.py"),
             os.path.join(py, "lib", "utils", "blob.py"), os.path.join(py, "lib", "utils", "timer.py")]
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n"] + files,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sub(os.path.join(py, "lib", "detect", "test.py"),
        [("if dets == []:", "if isinstance(dets, list) and dets == []:")])
    cfgp = os.path.join(py, "lib", "detect", "config.py")
    src = open(cfgp).read().expandtabs(8)
    open(cfgp, "w").write(src)
    stubs = os.path.join(tmp, "stubs")
    os.makedirs(stubs)
    open(os.path.join(stubs, "caffe.py"), "w").close()
    with open(os.path.join(stubs, "easydict.py"), "w") as f:
        f.write("class EasyDict(dict):\n"
                "    def __init__(self, d=None, **kw):\n"
                "        d = dict(d or {}); d.update(kw)\n"
                "        for k, v in d.items(): setattr(self, k, v)\n"
                "    def __setattr__(self, k, v):\n"
                "        if isinstance(v, dict) and not isinstance(v, EasyDict): v = EasyDict(v)\n"
                "        dict.__setitem__(self, k, v); dict.__setattr__(self, k, v)\n"
                "    __setitem__ = __setattr__\n"
                "    def has_key(self, k): return k in self\n")
    with open(os.path.join(stubs, "cv2.py"), "w") as f:
        f.write("import numpy as np\nINTER_LINEAR = 1\n"
                "def resize(im, a, b, fx=1.0, fy=1.0, interpolation=1):\n"
                "    h = int(round(im.shape[0] * fy)); w = int(round(im.shape[1] * fx))\n"
                "    return np.zeros((h, w) + im.shape[2:], dtype=im.dtype)\n")
    np.float = float
    np.int = int
    np.bool = bool
    sys.path.insert(0, stubs)
    # 'utils' resolves to the reference copy; an empty 'detect' package avoids
    # detect/__init__.py importing the training modules.
    sys.path.insert(0, os.path.join(py, "lib"))
    pkg = types.ModuleType("detect")
    pkg.__path__ = [os.path.join(py, "lib", "detect")]
    sys.modules["detect"] = pkg
    import detect.config as C
    import detect.test as T
    import utils.cython_div as cdiv
    import utils.cython_nms as cnms
    import utils.cython_bbox as cbbox
    assert cdiv.__file__.startswith(tmp) and T.__file__.startswith(tmp)
    return cdiv, cnms, cbbox, T, C


# ----------------------------------------------------------------------------
class RecordingNet(orc.OracleNet):
    """OracleNet that records every forward's rois and outputs."""

    def __init__(self, *a, **kw):
        orc.OracleNet.__init__(self, *a, **kw)
        self.rec = []

    def forward(self, blobs=None, **kw):
        out = orc.OracleNet.forward(self, blobs=blobs, **kw)
        self.rec.append({"rois": kw["rois"].copy(), "zoom_prob": out["zoom_prob"].copy(),
                         "adj_prob": out["adj_prob"].copy(), "adj_bbox": out["adj_bbox"].copy(),
                         "full": "data" in kw})
        return out


def expand_root(div, H, W, K):
    B = np.array([[0, 0, W - 1.0, H - 1.0]])
    ins, outs = [], []
    for _ in range(1, K):
        nxt = div.divide_region(B, 10.0)
        ins.append(B)
        outs.append(nxt)
        B = nxt
    return ins, outs


def pack_list(prefix, arrs, d):
    d[prefix + "_n"] = np.array([a.shape[0] for a in arrs], dtype=np.int64)
    d[prefix] = np.concatenate(arrs, axis=0) if arrs else np.zeros((0, 4))


def main():
    os.makedirs(GOLD, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="azref_")
    try:
        cdiv, cnms, cbbox, T, C = build_reference(tmp)
        rng = np.random.RandomState(20240)

        # ---------------- G1 divide_region ------------------------------------
        g = {}
        sizes = [(600, 1000), (375, 500), (480, 640), (800, 1200), (640, 853), (600, 600), (333, 500)]
        g["sizes"] = np.array(sizes)
        for i, (H, W) in enumerate(sizes):
            K = int(np.log2(min(H, W) // 10) + 1.0)
            ins, outs = expand_root(cdiv, H, W, K)
            pack_list("root%d_in" % i, ins, g)
            pack_list("root%d_out" % i, outs, g)
        x1 = rng.uniform(0, 900, 200)
        y1 = rng.uniform(0, 500, 200)
        w = rng.uniform(10, 300, 200)
        h = rng.uniform(10, 300, 200)
        rand = np.stack([x1, y1, x1 + w, y1 + h], 1)
        rand[:20, 2] = rand[:20, 0] + np.floor(w[:20])             # integer sides
        rand[:20, 3] = rand[:20, 1] + np.floor(w[:20])             # exact squares -> tie picks width
        rand[20:30, 3] = rand[20:30, 1] + 9.0 + 1e-9               # just above MIN_SIDE
        rand[30:40, 2] = rand[30:40, 0] + 12 * h[30:40]            # extreme aspect ratios
        rand[40:50, 3] = rand[40:50, 1] + 9 * w[40:50]
        g["rand_in"] = rand
        g["rand_out"] = cdiv.divide_region(rand, 10.0)
        singles = [cdiv.divide_region(rand[i:i + 1], 10.0) for i in range(60)]
        pack_list("single_out", singles, g)
        np.savez_compressed(os.path.join(GOLD, "g1_divide_region.npz"), **g)

        # ---------------- G2 _sift_dup ----------------------------------------
        g = {}
        base = rng.uniform(0, 1000, (300, 4))
        dup = np.vstack([base, base[rng.randint(0, 300, 200)] + rng.uniform(-3, 3, (200, 4))])
        half = np.round(rng.uniform(0, 100, (200, 4))) * 10.0 + 5.0      # x/10 ends in .5
        half[::3] += 1e-9
        half[1::3] -= 1e-9
        mix = np.vstack([dup, half, half[::-1]])
        g["in"] = mix
        g["out10"] = cdiv._sift_dup(mix, 10.0)
        g["out16"] = cdiv._sift_dup(mix, 16.0)
        np.savez_compressed(os.path.join(GOLD, "g2_sift_dup.npz"), **g)

        # ---------------- G4 decode / clip / unwrap ---------------------------
        g = {}
        C.cfg_set_mode("Test", 0.0)
        boxes = np.stack([x1, y1, x1 + w, y1 + h], 1)
        deltas = rng.uniform(-1, 1, (200, 44)).astype(np.float32)
        deltas[:10, 2::4] = -3.5            # shrink below MIN_SIDE
        deltas[10:20, 0::4] = 4.0           # push outside the image -> clipped
        scores = rng.uniform(0, 1, (200, 11)).astype(np.float32)
        pred = T._bbox_pred(boxes, deltas)
        g["boxes"], g["deltas"], g["scores"] = boxes, deltas, scores
        g["pred"] = pred.copy()
        clipped = T._clip_boxes(pred.copy(), (600, 1000, 3))
        g["clipped"] = clipped.copy()
        a, c = T._unwrap_adj_pred(clipped, scores)
        g["unwrap_boxes"], g["unwrap_scores"] = a, c
        np.savez_compressed(os.path.join(GOLD, "g4_decode.npz"), **g)

        # ---------------- G5 nms ----------------------------------------------
        g = {}
        cases = []
        for N in (1, 2, 100, 300, 2000, 8129):
            bx = rng.uniform(0, 900, N)
            by = rng.uniform(0, 500, N)
            bw = rng.uniform(10, 210, N)
            bh = rng.uniform(10, 210, N)
            sc = rng.permutation(N).astype(np.float64) / N + 1e-3         # distinct
            dets = np.stack([bx, by, bx + bw, by + bh, sc], 1).astype(np.float32)
            assert len(np.unique(dets[:, 4])) == N
            for t in (0.3, 0.5, 0.7):
                cases.append((dets, t))
        # integer boxes engineered so IoU == thresh exactly: pins `>=` (nms.pyx:64)
        eng = np.array([[0, 0, 9, 9, 0.9],       # area 100
                        [0, 0, 9, 4, 0.8],       # inter 50, union 100 -> 0.5 exactly
                        [100, 100, 109, 109, 0.7],
                        [100, 105, 109, 114, 0.6],   # inter 50, union 150 -> 1/3
                        [200, 200, 209, 209, 0.5],
                        [200, 200, 209, 206, 0.4]],  # inter 70, union 100 -> 0.7 (f32: 0.69999999)
                       dtype=np.float32)
        for t in (0.5, 1.0 / 3.0, float(np.float32(1.0) / np.float32(3.0)), 0.7,
                  float(np.float32(0.7)), 0.25):
            cases.append((eng, t))
        g["ncases"] = np.array(len(cases))
        for i, (dets, t) in enumerate(cases):
            g["dets%d" % i] = dets
            g["thresh%d" % i] = np.array(t)
            g["keep%d" % i] = np.array(cnms.nms(dets, t), dtype=np.int64)
        np.savez_compressed(os.path.join(GOLD, "g5_nms.npz"), **g)

        # ---------------- G6 bbox_overlaps ------------------------------------
        g = {}
        a = np.stack([x1[:40], y1[:40], x1[:40] + w[:40], y1[:40] + h[:40]], 1)
        q = np.stack([x1[40:60], y1[40:60], x1[40:60] + w[40:60], y1[40:60] + h[40:60]], 1)
        g["boxes"], g["query"] = a, q
        g["overlaps"] = cbbox.bbox_overlaps(a, q)
        np.savez_compressed(os.path.join(GOLD, "g6_bbox_overlaps.npz"), **g)

        # ---------------- G7/G3/G8 whole-loop traces with a recorded net ------
        head = synth.make_head(seed=77, **synth.SMALL_DIMS)
        traces = [  # (H, W, Tz quantile or value, BATCH_SIZE, MAX_SIZE)
            ("a", 600, 1000, 0.0, 10000, 1000),
            ("b", 375, 500, "q55", 10000, 1000),
            ("c", 480, 640, "q40", 10000, 1000),
            ("d", 640, 853, "q50", 100, 1000),       # chunked forward (test.py:195-205)
            ("e", 375, 500, 0.0, 1000, 800),         # voc.yml: MAX_SIZE 800, BATCH 1000
            ("f", 600, 1000, 1.5, 10000, 1000),      # nothing but the root is zoomed
        ]
        for tag, H, W, tz, batch, max_size in traces:
            C.cfg.TEST.MAX_SIZE = max_size
            C.cfg.SEAR.BATCH_SIZE = batch
            im = synth.make_image(3, H, W)
            # the reference computes the scale itself (test.py:45-50)
            scale = 600.0 / min(H, W)
            if np.round(scale * max(H, W)) > max_size:
                scale = float(max_size) / max(H, W)
            fh = synth.conv_out_size(int(round(H * scale)))
            fw = synth.conv_out_size(int(round(W * scale)))
            fmap = synth.make_feature_map(5, synth.SMALL_DIMS["C"], fh, fw)

            def run(Tz):
                C.cfg_set_mode("Test", Tz)
                full = RecordingNet(head, feat_fn=lambda data: fmap)
                fcn = RecordingNet(head)
                Y, conv = T.im_propose({"full": full, "fc": fcn}, im, return_conv=True)
                return Y, full.rec + fcn.rec

            if isinstance(tz, str):
                _, rec0 = run(0.0)
                allz = np.concatenate([r["zoom_prob"].ravel() for r in rec0])
                Tz = float(np.quantile(allz.astype(np.float64), int(tz[1:]) / 100.0))
                # keep clear of any zoom score so BLAS ulps cannot flip a decision
                gaps = np.abs(allz.astype(np.float64) - Tz)
                Tz += 0.0 if gaps.min() > 1e-4 else 2.5e-4
            else:
                Tz = float(tz)
            Y, rec = run(Tz)
            g = {"H": np.array(H), "W": np.array(W), "Tz": np.array(Tz), "batch": np.array(batch),
                 "max_size": np.array(max_size), "scale": np.array(scale), "Y": Y,
                 "fmap_shape": np.array(fmap.shape), "ncalls": np.array(len(rec))}
            for i, r in enumerate(rec):
                for k in ("rois", "zoom_prob", "adj_prob", "adj_bbox"):
                    g["c%d_%s" % (i, k)] = r[k]
                g["c%d_full" % i] = np.array(r["full"])
            np.savez_compressed(os.path.join(GOLD, "g7_trace_%s.npz" % tag), **g)
            print("trace", tag, (H, W), "Tz=%.6f" % Tz, "calls", [r["rois"].shape[0] for r in rec],
                  "Y", Y.shape)
        C.cfg.TEST.MAX_SIZE = 1000
        C.cfg.SEAR.BATCH_SIZE = 10000

        # ---------------- G9 im_detect_shared / _frcnn_forward (config 3) --------------------------
        class RecDet(orc.OracleDetNet):
            def __init__(self, *a, **kw):
                orc.OracleDetNet.__init__(self, *a, **kw)
                self.rec = []

            def forward(self, blobs=None, **kw):
                out = orc.OracleDetNet.forward(self, blobs=blobs, **kw)
                self.rec.append({"rois": kw["rois"].copy(), "cls_prob": out["cls_prob"].copy(),
                                 "bbox_pred": out["bbox_pred"].copy()})
                return out

        dhead = synth.make_det_head(seed=99, **synth.SMALL_DET_DIMS)
        for tag, H, W, tzv, batch in (("a", 375, 500, 0.6, 10000), ("b", 600, 1000, 0.0, 100)):
            C.cfg.SEAR.BATCH_SIZE = batch
            C.cfg_set_mode("Test", tzv)
            im = synth.make_image(4, H, W)
            scale = 600.0 / min(H, W)
            if np.round(scale * max(H, W)) > 1000:
                scale = 1000.0 / max(H, W)
            fmap = synth.make_feature_map(8, synth.SMALL_DIMS["C"], synth.conv_out_size(int(round(H * scale))),
                                          synth.conv_out_size(int(round(W * scale))))
            full = RecordingNet(head, feat_fn=lambda data: fmap)
            fcn = RecordingNet(head)
            det = RecDet(dhead)
            scores, pboxes = T.im_detect_shared({"full": full, "fc": fcn}, {"fc": det}, im, 21)
            g = {"H": np.array(H), "W": np.array(W), "Tz": np.array(tzv), "batch": np.array(batch),
                 "scale": np.array(scale), "scores": scores, "pred_boxes": pboxes,
                 "ndet": np.array(len(det.rec)), "fmap_seed": np.array(8)}
            # the proposals fed to the detector = union of the recorded det rois is not enough (dedup);
            # re-run the proposal step alone to record them
            full2 = RecordingNet(head, feat_fn=lambda data: fmap)
            fcn2 = RecordingNet(head)
            g["proposals"] = T.im_propose({"full": full2, "fc": fcn2}, im)
            for i, r in enumerate(det.rec):
                for k in ("rois", "cls_prob", "bbox_pred"):
                    g["d%d_%s" % (i, k)] = r[k]
            np.savez_compressed(os.path.join(GOLD, "g9_detect_%s.npz" % tag), **g)
            print("detect", tag, scores.shape, pboxes.shape, "det calls", [r["rois"].shape[0] for r in det.rec])
        C.cfg.SEAR.BATCH_SIZE = 10000
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()
