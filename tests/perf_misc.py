#!/usr/bin/env python3
"""Timings of the pieces around the proposal loop, GPU (through the C ABI) next to the CPU oracle:
NMS at the sizes of SURVEY 8(d), the Fast R-CNN head on the shared map (BASELINE config 3), the
deep-tree search (config 4), recall matching and the tuner's search.  Checker script (imports the
oracle, so it lives under tests/; not collected by pytest).  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "az-net_amd", "lib"))
sys.path.insert(0, os.path.join(HERE, ".."))
from aznet_hip import ffi, synth              # noqa: E402
from aznet_hip.net import HipAZNet, HipDetNet  # noqa: E402
from oracle import az_oracle as orc           # noqa: E402


def best(f, n=20, warm=3):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def main():
    out = {}
    head = synth.make_head(seed=1234, **synth.FULL_DIMS)
    net = HipAZNet(head, max_regions=4096)
    ctx = net.ctx
    rng = np.random.RandomState(0)
    # ---- NMS (lib/utils/nms.pyx), uniform boxes, distinct scores, thresh 0.5 -------------------
    nms = {}
    for n in (100, 300, 2000, 8129):
        x1 = rng.uniform(0, 900, n); y1 = rng.uniform(0, 500, n)
        dets = np.stack([x1, y1, x1 + rng.uniform(10, 210, n), y1 + rng.uniform(10, 210, n),
                         rng.permutation(n) / float(n)], 1).astype(np.float32)
        g = best(lambda: ctx.nms(dets, 0.5))
        c = best(lambda: orc.nms(dets, 0.5), n=5, warm=1)
        assert list(ctx.nms(dets, 0.5)) == list(orc.nms(dets, 0.5))
        nms[str(n)] = {"gpu_ms_incl_copies": round(g, 4), "cpu_oracle_ms": round(c, 4), "kept": len(orc.nms(dets, 0.5))}
    out["nms"] = nms
    # ---- deep tree: 800x1200 original (scale 0.75 -> 600x900), K = 7, Tz = 0 ---------------------
    fmap = synth.make_feature_map(4, 512, synth.conv_out_size(600), synth.conv_out_size(900))
    net.set_conv(fmap)
    p = ffi.AzContext.make_params(800, 1200, 0.75, 0.0)
    Y, st = net.propose(p, want_stats=True)
    ms = best(lambda: net.propose(p), n=30)
    out["deep_tree_800x1200"] = {"ms_per_image": round(ms, 3), "proposals_per_s": round(300e3 / ms),
                                 "regions_per_level": [int(st.level_regions[l]) for l in range(st.n_levels)],
                                 "unique_per_level": [int(st.level_unique[l]) for l in range(st.n_levels)],
                                 "candidates": int(st.n_candidates)}
    # ---- Fast R-CNN head on the shared map: 300 proposals of a 600x1000 image ----------------------
    fmap = synth.make_feature_map(4, 512, 38, 63)
    net.set_conv(fmap)
    p = ffi.AzContext.make_params(600, 1000, 1.0, 0.0)
    Y = net.propose(p)
    det = HipDetNet(synth.make_det_head(seed=99), net)
    ms_det = best(lambda: det.detect(Y, 1.0, (600, 1000), 1. / 16., 10000, 1e-14), n=30)
    ms_prop = best(lambda: net.propose(p), n=30)
    out["shared_detection_600x1000"] = {"az_propose_ms": round(ms_prop, 3), "az_detect_300_rois_ms": round(ms_det, 3),
                                        "images_per_s": round(1e3 / (ms_prop + ms_det), 1)}
    # ---- tuner search (one level more, anchors recorded) ---------------------------------------------
    pt = ffi.AzContext.make_params(600, 1000, 1.0, 0.0, num_proposals=2000, tune=True)
    net.propose(pt)
    out["tuner_search_600x1000"] = {"ms_per_image": round(best(lambda: net.propose(pt), n=20), 3),
                                    "anchors": int(ctx.last_anchors()[0].shape[0])}
    # ---- recall matching: 4952 images x 300 proposals (VOC07 test size) -----------------------------
    cands, gts = [], []
    for i in range(4952):
        k = int(rng.randint(1, 8))
        gt = np.floor(rng.uniform(0, 400, (k, 4))); gt[:, 2:] += gt[:, :2] + 10
        b = rng.uniform(0, 400, (300, 4)); b[:, 2:] += b[:, :2] + 10
        b[:k] = gt + rng.uniform(-6, 6, (k, 4))
        cands.append(b); gts.append(gt)
    g = best(lambda: ctx.recall_match(cands, gts), n=5, warm=1)
    t0 = time.perf_counter()
    ref = orc.recall_gt_overlaps(cands[:500], gts[:500])
    c = (time.perf_counter() - t0) * 1e3 * 4952 / 500.0
    assert np.array_equal(ctx.recall_match(cands[:500], gts[:500]), ref)
    out["recall_match_4952x300"] = {"gpu_ms_incl_packing_and_copies": round(g, 2), "cpu_oracle_ms_extrapolated": round(c, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
