#!/usr/bin/env python3
"""Builder's probe (not a test): what a STREAM of distinct images costs at a threshold tuned over the set -- trees per image,
search forms, reruns, ms per image -- next to the same-tree replay of a few of its images.
    python tests/dev/stream_probe.py [--images 32] [--anchors 20,100,400] [--lanes 2] [--passes 3]"""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "az-net_amd", "lib"))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=32)
    ap.add_argument("--anchors", default="20,100,400")
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--noise", action="store_true", help="uniform-noise images (synth.make_image) instead of scenes")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--objects", action="store_true", help="planted-object maps + object head instead of images through the backbone")
    ap.add_argument("--depth", type=int, default=0)
    ap.add_argument("--replay", type=int, default=4, help="images whose same-tree replay is timed for comparison")
    ap.add_argument("--batch-only", action="store_true", help="skip the one-image-at-a-time stream and the replays")
    ap.add_argument("--batch", default="", help="also: the set in lockstep batches of these sizes (az_batch_launch), e.g. 4,8,16")
    args = ap.parse_args()
    import torch
    from aznet_hip import ffi, synth
    from aznet_hip.net import HipAZNet
    from aznet_hip.backbone import VGG16Conv5
    from detect.test import _get_image_blob
    H, W = 600, 1000
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = True
    if args.objects:
        head = synth.make_object_head(seed=1234, **synth.FULL_DIMS)
        net = HipAZNet(head, name="probe", max_regions=4096)
        net.ctx.set_lanes(args.lanes)
        convs = [torch.from_numpy(synth.make_object_map(j, 512, 38, 63)).cuda().contiguous(memory_format=torch.channels_last)
                 for j in range(args.images)]
    else:
        head = synth.make_head(seed=1234, **synth.FULL_DIMS)
        backbone = VGG16Conv5(device=dev, seed=4321, channels_last_out=True, channels_last_compute=True)
        net = HipAZNet(head, backbone=backbone, device=0, name="probe", max_regions=4096)
        net.ctx.set_lanes(args.lanes)
        mk_im = synth.make_image if args.noise else synth.make_scene_image
        ims = [mk_im(j, H, W) for j in range(args.images)]
        blob0, scales = _get_image_blob(synth.make_image(0, H, W), net)
        backbone.normalize_output(blob0)
        convs = [net.compute_conv(_get_image_blob(x, net)[0]).clone().contiguous(memory_format=torch.channels_last) for x in ims]
    print("map rms:", ["%.2f" % float(c.pow(2).mean().sqrt()) for c in convs[:8]])
    # the tuner's pool over the set (detect.tune.tune_thresh)
    net.ctx.tune_begin(args.images * 2 * net.ctx.max_regions)
    per_level = []
    for c in convs:
        net.set_conv(c)
        _, st = net.propose(ffi.AzContext.make_params(H, W, 1.0, 0.0, tune=True), want_stats=True)
    for a in [int(x) for x in args.anchors.split(",")]:
        tz, npool = net.ctx.tune_kth_largest(args.images * a)
        print("anchors/img %d -> Tz %.6f (pool %d)" % (a, tz, npool))
        per_level.append((a, tz))
    net.ctx.tune_end()
    depth = args.depth or args.lanes + 1

    def stream(prm, seq, stats=None):
        launched = 0
        for i in range(len(seq)):
            while launched < min(len(seq), i + depth):
                net.ctx.propose_launch(prm, fmap=convs[seq[launched]], producer_done=True)
                launched += 1
            Y, st = net.ctx.propose_fetch(want_stats=True)
            if stats is not None:
                stats.append((seq[i], st))

    import gc
    gc.collect()
    gc.disable()
    for a, tz in per_level:
        prm = ffi.AzContext.make_params(H, W, 1.0, tz)
        order = list(range(args.images))
        stream(prm, order)                                # the dataset's first pass (untimed: plans, histories)
        stats = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(1 if args.batch_only else args.passes):
            stream(prm, order, stats)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n = len(stats)
        forms = {}
        reruns = 0
        for _, st in stats:
            forms[ffi.SEARCH_FORMS[int(st.search_form)]] = forms.get(ffi.SEARCH_FORMS[int(st.search_form)], 0) + 1
            reruns += int(st.n_reruns)
        print("== anchors/img %d Tz %.5f: %.4f ms/image over %d images, reruns %d, forms %s" % (a, tz, dt / n * 1e3, n, reruns, forms))
        for bs in [int(x) for x in args.batch.split(",") if x]:
            groups = [order[i:i + bs] for i in range(0, len(order), bs)]
            inflight = 2 * args.lanes

            def run_batches(collect=None):
                launched = 0
                for gi in range(len(groups)):
                    while launched < min(len(groups), gi + inflight):
                        net.ctx.batch_launch(prm, [convs[j] for j in groups[launched]], producer_done=True)
                        launched += 1
                    rs = net.ctx.batch_fetch_all(want_stats=collect is not None)
                    if collect is not None:
                        collect.extend(r[1] for r in rs)
            for _ in range(max(2, -(-2 * inflight // len(groups)) + 1)):      # (every lane's two slot sets created, hints set)
                run_batches()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.passes):
                run_batches()
            torch.cuda.synchronize()
            dtb = time.perf_counter() - t0
            sts = []
            run_batches(sts)
            print("   lockstep batches of %2d (%d in flight): %.4f ms/image; reruns %d, forms %s" % (
                bs, inflight, dtb / (args.passes * len(order)) * 1e3, sum(int(s.n_reruns) for s in sts),
                sorted({int(s.search_form) for s in sts})))
        for i, st in stats[-args.images:] if args.verbose else []:
            print("   img %2d regions %s passes %s form %d reruns %d deferred %d" % (
                i, [int(st.level_regions[l]) for l in range(st.n_levels)], [int(x) for x in list(st.pass_rows)[:int(st.n_passes)]],
                int(st.search_form), int(st.n_reruns), int(st.root_deferred)))
        if args.batch_only:
            continue
        # same-tree replay of a few images (history primed with the image's own tree)
        rep = []
        for i in range(min(args.replay, args.images)):
            for _ in range(4):
                stream(prm, [i] * 4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sti = []
            stream(prm, [i] * 40, sti)
            torch.cuda.synchronize()
            rep.append((time.perf_counter() - t0) / 40 * 1e3)
            if args.verbose:
                st = sti[-1][1]
                print("   replay img %2d %.3f ms regions %s passes %s form %d deferred %d" % (
                    i, rep[-1], [int(st.level_regions[l]) for l in range(st.n_levels)], [int(x) for x in list(st.pass_rows)[:int(st.n_passes)]],
                    int(st.search_form), int(st.root_deferred)))
        print("   same-tree replay of images 0..%d: %s ms/image (mean %.4f); stream / replay = %.3f" % (
            len(rep) - 1, ["%.3f" % x for x in rep], float(np.mean(rep)), (dt / n * 1e3) / float(np.mean(rep))))
        trees = {}
        for i, st in stats[:args.images]:
            k = tuple(int(st.level_regions[l]) for l in range(st.n_levels))
            trees[k] = trees.get(k, 0) + 1
        print("   trees:", sorted(trees.items(), key=lambda kv: -kv[1])[:12])
        pk = {}
        for i, st in stats:
            k = (tuple(int(st.level_regions[l]) > 0 for l in range(st.n_levels)).count(True), tuple(int(x) > 0 for x in list(st.pass_rows)[:int(st.n_passes)]).count(True), int(st.n_reruns))
            pk[k] = pk.get(k, 0) + 1
        print("   (levels reached, passes, reruns): count", sorted(pk.items()))
        # the stream restricted to those images, for a like-for-like ratio
        sub = list(range(min(args.replay, args.images)))
        stream(prm, sub * 4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stream(prm, sub * 25)
        torch.cuda.synchronize()
        print("   those images as a stream: %.4f ms/image" % ((time.perf_counter() - t0) / (25 * len(sub)) * 1e3))


if __name__ == "__main__":
    main()
