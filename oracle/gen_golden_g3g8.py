#!/usr/bin/env python3
"""Golden vectors G3 and G8 of SURVEY 8(c), produced by the REFERENCE's own code (same recipe as gen_golden.py: the reference
is copied to a temp dir, made importable, run on seeded inputs; only arrays are written to tests/golden/).

Both computations are inline statements of lib/detect/test.py, not functions:
  G3  the feature-space dedup of `_az_forward` (test.py:210-218): `np.unique(hashes, return_index, return_inverse)`
  G8  the final selection of `im_propose` (test.py:397-401): `np.argsort(-aScores)`
so the reference's functions are run whole and the module's `np` is replaced by a spy that records the arguments and results
of exactly those two calls -- what the reference itself computed, not a restatement of it.

  g3_roi_dedup.npz   every level of the full trees of three image shapes whose test scales are 1.0 (600x1000), 1.6 (375x500)
                     and 0.9375 (640x853), plus chunked levels (BATCH_SIZE 100): boxes, scale, hashes, index, inv_index
  g8_topk.npz        whole im_propose runs (recorded net) whose adjacency scores are (a) distinct, (b) quantised so that many
                     candidates tie: -aScores as handed to argsort, indA, the candidate boxes, Y

Run:  python oracle/gen_golden_g3g8.py      (this container only; needs /root/reference)
"""
import os
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


class NpSpy(object):
    """Stands in for the `np` global of the reference's detect.test: everything is NumPy's, `unique` and `argsort` also
    keep what they were given and what they returned."""

    def __init__(self, real):
        self._real = real
        self.unique_calls = []
        self.argsort_calls = []

    def __getattr__(self, k):
        return getattr(self._real, k)

    def unique(self, a, **kw):
        out = self._real.unique(a, **kw)
        self.unique_calls.append((self._real.array(a, copy=True), out))
        return out

    def argsort(self, a, *args, **kw):
        out = self._real.argsort(a, *args, **kw)
        self.argsort_calls.append((self._real.array(a, copy=True), out.copy()))
        return out


class ZeroNet(object):
    """A net whose outputs do not matter (G3 records what happens BEFORE the forward)."""

    def __init__(self, fmap_shape):
        self.name = "zero"
        self.blobs = {k: orc._Blob() for k in ("data", "rois", "conv5_3")} if hasattr(orc, "_Blob") else None
        if self.blobs is None:
            class B(object):
                def reshape(self, *s):
                    self.shape = s
            self.blobs = {k: B() for k in ("data", "rois", "conv5_3")}
        self.fmap = np.zeros(fmap_shape, dtype=np.float32)

    def keys(self):
        return ["full", "fc"]

    def forward(self, blobs=None, **kw):
        R = kw["rois"].shape[0]
        out = {"zoom_prob": np.zeros((R, 1), np.float32), "adj_prob": np.zeros((R, 11), np.float32),
               "adj_bbox": np.zeros((R, 44), np.float32)}
        for b in blobs or []:
            out[b] = self.fmap
        return out


class QuantNet(gg.RecordingNet):
    """RecordingNet whose adjacency scores are rounded to a coarse grid: many candidates share a score."""

    def __init__(self, *a, **kw):
        self.steps = kw.pop("steps")
        gg.RecordingNet.__init__(self, *a, **kw)

    def forward(self, blobs=None, **kw):
        out = orc.OracleNet.forward(self, blobs=blobs, **kw)
        out["adj_prob"] = (np.round(out["adj_prob"] * self.steps) / self.steps).astype(np.float32)
        self.rec.append({"rois": kw["rois"].copy(), "zoom_prob": out["zoom_prob"].copy(),
                         "adj_prob": out["adj_prob"].copy(), "adj_bbox": out["adj_bbox"].copy(), "full": "data" in kw})
        return out


def main():
    os.makedirs(GOLD, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="azref_")
    try:
        cdiv, cnms, cbbox, T, C = gg.build_reference(tmp)
        spy = NpSpy(np)
        T.np = spy
        # ---------------- G3 feature-space dedup (test.py:210-218) ---------------------------------------------------
        g = {}
        cases = []
        C.cfg_set_mode("Test", 0.0)
        for (H, W), batch in (((600, 1000), 10000), ((375, 500), 10000), ((640, 853), 10000), ((375, 500), 100)):
            C.cfg.SEAR.BATCH_SIZE = batch
            C.cfg.TEST.MAX_SIZE = 1000
            scale = 600.0 / min(H, W)
            assert np.round(scale * max(H, W)) <= 1000
            K = int(np.log2(min(H, W) // 10) + 1.0)
            ins, _ = gg.expand_root(cdiv, H, W, K)
            im = np.zeros((H, W, 3), dtype=np.uint8)
            net = ZeroNet((1, 8, 4, 4))
            for B in ins:
                del spy.unique_calls[:]
                T._az_forward({"full": net, "fc": net}, im, B, None)
                nb = int(np.ceil(B.shape[0] / float(batch)))
                assert len(spy.unique_calls) == nb
                for bid, (hashes, (_, index, inv)) in enumerate(spy.unique_calls):
                    cases.append({"boxes": B[batch * bid:min(B.shape[0], batch * (bid + 1))].copy(), "scale": scale,
                                  "hashes": hashes, "index": np.asarray(index, dtype=np.int64),
                                  "inv_index": np.asarray(inv, dtype=np.int64).ravel(), "H": H, "W": W, "batch": batch})
        C.cfg.SEAR.BATCH_SIZE = 10000
        g["ncases"] = np.array(len(cases))
        for i, c in enumerate(cases):
            for k, v in c.items():
                g["c%d_%s" % (i, k)] = np.asarray(v)
        np.savez_compressed(os.path.join(GOLD, "g3_roi_dedup.npz"), **g)
        print("g3:", len(cases), "dedup calls; scales", sorted({float(c["scale"]) for c in cases}),
              "rows", [c["boxes"].shape[0] for c in cases], "unique", [c["index"].shape[0] for c in cases])

        # ---------------- G8 final top-K (test.py:397-401) -----------------------------------------------------------
        g = {}
        head = synth.make_head(seed=77, **synth.SMALL_DIMS)
        runs = []
        for tag, H, W, steps, nprop in (("distinct", 375, 500, 0, 300), ("ties", 375, 500, 64, 300),
                                        ("ties_coarse", 480, 640, 8, 300), ("short", 600, 1000, 0, 4000)):
            C.cfg_set_mode("Test", 0.0)
            C.cfg.SEAR.NUM_PROPOSALS = nprop
            im = synth.make_image(6, H, W)
            scale = 600.0 / min(H, W)
            fmap = synth.make_feature_map(9, synth.SMALL_DIMS["C"], synth.conv_out_size(int(round(H * scale))),
                                          synth.conv_out_size(int(round(W * scale))))
            if steps:
                full, fcn = QuantNet(head, feat_fn=lambda d: fmap, steps=steps), QuantNet(head, steps=steps)
            else:
                full, fcn = gg.RecordingNet(head, feat_fn=lambda d: fmap), gg.RecordingNet(head)
            del spy.argsort_calls[:]
            Y = T.im_propose({"full": full, "fc": fcn}, im)
            assert len(spy.argsort_calls) == 1
            neg, indA = spy.argsort_calls[0]
            # the candidate boxes in the reference's order: the same run once more with the selection switched to "all"
            C.cfg.SEAR.NUM_PROPOSALS = 10 ** 9
            del spy.argsort_calls[:]
            full.rec, fcn.rec = [], []
            Yfull = T.im_propose({"full": full, "fc": fcn}, im)
            neg2, indA2 = spy.argsort_calls[0]
            assert np.array_equal(neg, neg2) and np.array_equal(indA, indA2)
            Yall = np.empty_like(Yfull)
            Yall[indA2] = Yfull                      # Yfull = Y_all[indA]  ->  Y_all
            assert np.array_equal(Yall[indA[:Y.shape[0]]], Y)
            runs.append(tag)
            g[tag + "_neg_scores"] = neg
            g[tag + "_indA"] = np.asarray(indA, dtype=np.int64)
            g[tag + "_Y_all"] = Yall
            g[tag + "_Y"] = Y
            g[tag + "_num_proposals"] = np.array(nprop)
            nt = neg.size - np.unique(neg).size
            print("g8:", tag, "candidates", neg.size, "tied", nt, "Y", Y.shape)
        g["runs"] = np.array(runs)
        C.cfg.SEAR.NUM_PROPOSALS = 300
        np.savez_compressed(os.path.join(GOLD, "g8_topk.npz"), **g)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print("g3 / g8 written to", GOLD)


if __name__ == "__main__":
    main()
