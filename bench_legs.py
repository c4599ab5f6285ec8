#!/usr/bin/env python3
"""Side legs of bench.py that stand on their own (run only with `bench.py --extras`; their results go to bench_extras.json,
never into the printed line): `stream_tz` -- distinct images in dataset order at ONE threshold tuned over the set, one image
per search and in lockstep batches, images of several shapes -- and `extras` -- BASELINE config 4 (deep tree, both network
scales), config 3 (shared detection) and az_nms at SURVEY 8(d)'s sizes."""
import time

import numpy as np

from bench import H_IM, NUM_PROPOSALS, W_IM, floors, merged_floor_batches, t_min_us


def stream_tz(net, head, backbone, convs0, ffi, synth, HipAZNet, torch, get_image_blob, args, depth, device):
    """What tools/prop_az.py runs (prop_az.py:74-79 -> test.py:508-513): cfg_set_mode('Test', Tz) with ONE threshold tuned
    over an image set (detect.tune.tune_thresh: the score that ANCHORS_PER_IMG anchors per image exceed on average --
    az_tune_begin / az_tune_kth_largest), then image after image, every one a different tree.  No per-image priming: the
    context's history when image i is launched is what the images before it left.  Two sets of 600x1000 maps:
      objects    conv5_3 stand-ins with a few planted objects each + a head whose zoom unit reads them (synth.make_object_*):
                 the zoom indicator is high where an object is and consistent from a region to its sub-regions, as a TRAINED
                 AZ-Net's is -- deep, sparse trees that differ from image to image; tuned at cfg.TRAIN.ANCHORS_PER_IMG = 20
                 (config.py:104), the reference's own setting;
      untrained  scene images (synth.make_scene_image) through the random-weight backbone and head of `value`: the zoom
                 score drifts with region size, so one threshold gives a MIXTURE of trees that end at their second level and
                 dense ones -- the hardest case for choices made from history; at 20 and at 1500 anchors per image.
    Per point: the set once untimed (the dataset's first images: plans, histories), then `passes` x the set timed in order,
    queue-ahead as `value`; searches run twice; the forms taken; every image's own same-tree replay (history primed with
    its own tree, the round-4 measurement) for the ratio; the mean fraction of the images' merged-pass floors."""
    import gc
    H, W = H_IM, W_IM
    n_img = max(8, int(args.stream_images))
    passes = 3
    res = {"images_per_set": n_img, "timed_images_per_point": passes * n_img, "lanes": int(getattr(net.ctx, "lanes", 1)),
           "searches_launched_and_unfetched": depth, "points": []}

    def run_set(cnet, maps, prm, seq, stats=None):
        launched = 0
        for i in range(len(seq)):
            while launched < min(len(seq), i + depth):
                cnet.ctx.propose_launch(prm, fmap=maps[seq[launched]], producer_done=True)
                launched += 1
            Y, st = cnet.ctx.propose_fetch(want_stats=True)
            if stats is not None:
                stats.append((seq[i], st))

    def tune(cnet, maps, anchors_per_img):
        cnet.ctx.tune_begin(len(maps) * 2 * cnet.ctx.max_regions)
        for m in maps:
            cnet.set_conv(m)
            cnet.propose(ffi.AzContext.make_params(H, W, 1.0, 0.0, num_proposals=NUM_PROPOSALS, tune=True))
        tz = [cnet.ctx.tune_kth_largest(len(maps) * a)[0] for a in anchors_per_img]
        cnet.ctx.tune_end()
        return tz

    def point(label, cnet, maps, tz, anchors):
        prm = ffi.AzContext.make_params(H, W, 1.0, tz, num_proposals=NUM_PROPOSALS)
        fm = int(maps[0].numel())
        order = list(range(len(maps)))
        gc.collect()
        gc.disable()
        run_set(cnet, maps, prm, order)                       # the dataset's first pass: untimed
        stats = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(passes):
            run_set(cnet, maps, prm, order, stats)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n = len(stats)
        ms = dt / n * 1e3
        forms, reruns, fracs, trees = {}, 0, [], []
        for i, st in stats:
            f = ffi.SEARCH_FORMS.get(int(st.search_form), "?")
            forms[f] = forms.get(f, 0) + 1
            reruns += int(st.n_reruns)
        for i, st in stats[:len(maps)]:
            trees.append([int(st.level_regions[l]) for l in range(st.n_levels)])
        # every image's same-tree replay: 12 untimed searches of that image alone, then 30 timed
        rep = []
        for i in order:
            run_set(cnet, maps, prm, [i] * 12)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st_i = []
            run_set(cnet, maps, prm, [i] * 30, st_i)
            torch.cuda.synchronize()
            r_ms = (time.perf_counter() - t0) / 30 * 1e3
            rep.append(r_ms)
            fracs.append(floors(st_i[-1][1], fm, 1.0)["merged_pass_t_min_us"])
        # the same set in lockstep batches (az_batch_launch): the images of B consecutive iterations of the dataset loop walk
        # their trees together, every level's rois of all of them in ONE head pass; two batches in flight per lane
        lock = {}
        lanes = int(getattr(cnet.ctx, "lanes", 1))
        for bs in (4, 8, 16, 32):
            groups = [order[i:i + bs] for i in range(0, len(order), bs)]

            def run_batches(collect=None):
                launched = 0
                for gi in range(len(groups)):
                    while launched < min(len(groups), gi + 2 * lanes):
                        cnet.ctx.batch_launch(prm, [maps[j] for j in groups[launched]], producer_done=True)
                        launched += 1
                    rs = cnet.ctx.batch_fetch_all(want_stats=collect is not None)
                    if collect is not None:
                        collect.extend(r[1] for r in rs)
            # (untimed: every lane's two slot sets are created and have a batch's row counts behind them; the first size also
            #  takes the idle gap)
            for _ in range(max(4 if bs == 4 else 2, -(-4 * lanes // len(groups)) + 1)):
                run_batches()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(passes):
                run_batches()
            torch.cuda.synchronize()
            b_ms = (time.perf_counter() - t0) / (passes * len(order)) * 1e3
            sts = []
            run_batches(sts)
            bforms = {}
            for st in sts:
                f = ffi.SEARCH_FORMS.get(int(st.search_form), "?")
                bforms[f] = bforms.get(f, 0) + 1
            lock[str(bs)] = {"ms_per_image": b_ms, "value": NUM_PROPOSALS * 1e3 / b_ms, "vs_one_image_at_a_time": ms / b_ms,
                             "searches_run_twice": sum(int(st.n_reruns) for st in sts), "search_forms": bforms,
                             "rows_per_pass_mean": merged_floor_batches(sts, bs, fm)[1],
                             "merged_pass_floor_frac": merged_floor_batches(sts, bs, fm)[0] / (b_ms * 1e3)}
        gc.enable()
        merged_mean = float(np.mean(fracs))
        reg = np.array([t + [0] * (8 - len(t)) for t in trees])[:, :len(trees[0])]
        return {"set": label, "anchors_per_img": anchors, "Tz": tz, "ms_per_image": ms, "value": NUM_PROPOSALS * 1e3 / ms,
                "lockstep_batches": lock,
                "unit": "proposals/s", "timed_images": n, "searches_run_twice": reruns, "rerun_rate": reruns / float(n),
                "search_forms": forms,
                "same_tree_replay_ms_per_image": float(np.mean(rep)), "stream_over_replay": ms / float(np.mean(rep)),
                "merged_pass_floor": {"mean_t_min_us_per_image": merged_mean, "frac": merged_mean / (ms * 1e3),
                                      "frac_of_replay": merged_mean / (float(np.mean(rep)) * 1e3)},
                "regions_per_level": {"mean": [float(x) for x in reg.mean(0)], "min": [int(x) for x in reg.min(0)],
                                      "max": [int(x) for x in reg.max(0)]},
                "trees_of_the_first_images": trees[:8],
                "levels_reached_histogram": {str(k): int(v) for k, v in
                                             zip(*np.unique([sum(1 for x in t if x > 0) for t in trees], return_counts=True))}}

    def mixed_shapes_point(cnet, n):
        shapes = [(375, 500), (500, 375), (333, 500), (375, 500), (500, 333), (375, 500), (500, 375), (600, 1000)]
        items = []
        for j in range(n):
            h, w = shapes[j % len(shapes)]
            sc = 600.0 / min(h, w)
            if round(sc * max(h, w)) > 1000:
                sc = 1000.0 / max(h, w)
            fh, fw = synth.conv_out_size(int(round(h * sc))), synth.conv_out_size(int(round(w * sc)))
            m = torch.from_numpy(synth.make_object_map(500 + j, 512, fh, fw)).to("cuda:%d" % device).contiguous(memory_format=torch.channels_last)
            items.append((h, w, sc, m))
        cnet.ctx.tune_begin(n * 2 * cnet.ctx.max_regions)
        for (h, w, sc, m) in items:
            cnet.set_conv(m)
            cnet.propose(ffi.AzContext.make_params(h, w, sc, 0.0, num_proposals=NUM_PROPOSALS, tune=True))
        tz = cnet.ctx.tune_kth_largest(n * 20)[0]
        cnet.ctx.tune_end()
        prm = [ffi.AzContext.make_params(h, w, sc, tz, num_proposals=NUM_PROPOSALS) for (h, w, sc, _) in items]
        maps = [m for (_, _, _, m) in items]
        order = list(range(n))

        def one_by_one(stats=None):
            launched = 0
            for i in range(n):
                while launched < min(n, i + depth):
                    cnet.ctx.propose_launch(prm[launched], fmap=maps[launched], producer_done=True)
                    launched += 1
                Y, st = cnet.ctx.propose_fetch(want_stats=True)
                if stats is not None:
                    stats.append(st)
        gc.collect()
        gc.disable()
        one_by_one()
        one_by_one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(passes):
            one_by_one()
        torch.cuda.synchronize()
        ms1 = (time.perf_counter() - t0) / (passes * n) * 1e3
        lanes = int(getattr(cnet.ctx, "lanes", 1))
        lock = {}
        for bs in (8, 16):
            groups = [order[i:i + bs] for i in range(0, n, bs)]

            def run_batches(collect=None):
                launched = 0
                for gi in range(len(groups)):
                    while launched < min(len(groups), gi + 2 * lanes):
                        cnet.ctx.batch_launch([prm[j] for j in groups[launched]], [maps[j] for j in groups[launched]], producer_done=True)
                        launched += 1
                    rs = cnet.ctx.batch_fetch_all(want_stats=collect is not None)
                    if collect is not None:
                        collect.extend(r[1] for r in rs)
            for _ in range(max(2, -(-4 * lanes // len(groups)) + 1)):
                run_batches()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(passes):
                run_batches()
            torch.cuda.synchronize()
            b_ms = (time.perf_counter() - t0) / (passes * n) * 1e3
            sts = []
            run_batches(sts)
            lock[str(bs)] = {"ms_per_image": b_ms, "value": NUM_PROPOSALS * 1e3 / b_ms, "vs_one_image_at_a_time": ms1 / b_ms,
                             "searches_run_twice": sum(int(st.n_reruns) for st in sts),
                             "search_forms": sorted({ffi.SEARCH_FORMS.get(int(st.search_form), "?") for st in sts})}
        gc.enable()
        return {"set": "objects, images of %d shapes in dataset order (%s)" % (len(set(shapes)), ", ".join("%dx%d" % sh for sh in sorted(set(shapes)))),
                "anchors_per_img": 20, "Tz": tz, "images": n, "one_image_at_a_time_ms": ms1, "value": NUM_PROPOSALS * 1e3 / ms1,
                "unit": "proposals/s", "lockstep_batches": lock,
                "note": "every batch holds images of several shapes (az_batch_launch_shapes: per-image map size, pre-pass, clipping "
                        "box; shared head passes)"}

    # ---- objects: planted-object maps + a head whose zoom unit reads them ---------------------------------------------
    ohead = synth.make_object_head(seed=1234, **synth.FULL_DIMS)
    onet = HipAZNet(ohead, backbone=None, device=device, name="stream_objects", max_regions=4096)
    onet.ctx.set_lanes(int(getattr(net.ctx, "lanes", 1)))
    omaps = [torch.from_numpy(synth.make_object_map(j, 512, 38, 63)).to("cuda:%d" % device).contiguous(memory_format=torch.channels_last)
             for j in range(n_img)]
    tz_o = tune(onet, omaps, [20])
    res["points"].append(point("objects", onet, omaps, tz_o[0], 20))
    # ---- the same head over images of SEVERAL shapes (what a dataset is: VOC's 500x375, 375x500, 500x333 ... at the reference's
    #      scale rule), one image per search and in lockstep batches whose images differ in shape (az_batch_launch_shapes)
    try:
        res["mixed_shapes"] = mixed_shapes_point(onet, n_img)
    except Exception as e:                                   # noqa: BLE001 -- an extra, never the bench line's failure
        res["mixed_shapes"] = {"error": repr(e)}
    del onet, omaps, ohead
    # ---- untrained: scene images through `value`'s backbone and head --------------------------------------------------------
    smaps = []
    for j in range(n_img):
        b = get_image_blob(synth.make_scene_image(j, H, W), net)[0]
        smaps.append(net.compute_conv(b).clone().contiguous(memory_format=torch.channels_last))
    tz_s = tune(net, smaps, [20, 1500])
    for a, tz in zip([20, 1500], tz_s):
        res["points"].append(point("untrained", net, smaps, tz, a))
    res["note"] = ("one threshold per point, tuned over its set with az_tune_*; images launched in dataset order, no per-image "
                   "priming; rerun_rate = searches that had to be run twice (an early end that missed, a whole-tree pass that "
                   "lacked a window) / timed images; stream_over_replay = ms_per_image over the mean of every image's own "
                   "history-primed replay (what calibrated_tz / tz_sweep measure for ONE image); merged_pass_floor.frac = mean "
                   "of the images' merged-pass floors (of their replay's passes) over ms_per_image; lockstep_batches = the same "
                   "set, B consecutive images per az_batch_launch (each image's result identical to its search alone: "
                   "tests/test_gpu_batch.py), two batches in flight per lane; its floor charges one weight stream per pass of "
                   "a BATCH")
    return res


def extras(net, head, ffi, synth, HipDetNet, torch, args):
    """deep_tree (BASELINE config 4), shared_detection (config 3) and az_nms at SURVEY 8(d)'s sizes, each with the
    kernel time of its launches from HIP events on the ctx stream (az_set_profiling) next to the wall clock through
    the C ABI."""
    res = {}
    ctx = net.ctx

    def wall(f, n, warm=3):
        for _ in range(warm):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def kernel_ms(f, n):
        """sum of the HIP-event times of every launch group of one call, averaged over n calls"""
        ctx.set_profiling(0)
        ctx.set_profiling(2 | 4)
        for _ in range(n):
            f()
        kt = ctx.last_kernel_times()
        ctx.set_profiling(0)
        by = {}
        for nm, l, ms in kt:
            by[nm] = by.get(nm, 0.0) + ms / n
        return sum(by.values()), by

    # ---- config 4: 800x1200 original, K = 7 -- at scale 0.75 (the reference's rule: 600-px short side, conv5_3 38x57) and at
    #      scale 1.0 (an 800-px NETWORK short side, conv5_3 50x75: BASELINE config 4 read literally) --------------------
    n_img = max(10, min(40, args.steps // 5))

    def deep(scale, forms):
        out_d = None
        fmap = synth.make_feature_map(4, 512, synth.conv_out_size(int(round(800 * scale))), synth.conv_out_size(int(round(1200 * scale))))
        net.set_conv(fmap)
        for form, static in forms:
            p = ffi.AzContext.make_params(800, 1200, scale, 0.0, static_tree=static)
            Yd, std = net.propose(p, want_stats=True)
            ms = wall(lambda: net.propose(p), n_img)
            Yd, std = net.propose(p, want_stats=True)          # (the passes of a search that has the context's history)
            ud = [int(std.level_unique[l]) for l in range(std.n_levels)]
            fl = t_min_us(ud, int(fmap.size))
            kms, by = kernel_ms(lambda: net.propose(p), 5)
            # ... and as `value` is measured: searches queued ahead on the context's lanes (throughput; the figure above is one
            # image at a time, i.e. latency)
            tmap = torch.from_numpy(fmap).to("cuda:%d" % ctx.device).contiguous(memory_format=torch.channels_last)
            dq = int(getattr(ctx, "lanes", 1)) + 1

            def queued(k):
                launched = 0
                for i in range(k):
                    while launched < min(k, i + dq):
                        ctx.propose_launch(p, fmap=tmap, producer_done=True)
                        launched += 1
                    ctx.propose_fetch()
            queued(12)                                 # (both lanes' histories and plans for this shape settle: a plan built in
            torch.cuda.synchronize()                   #  the timed region showed up once as 8.8 ms per image in a 10-image sample)
            t0 = time.perf_counter()
            queued(n_img)
            torch.cuda.synchronize()
            ms_q = (time.perf_counter() - t0) / n_img * 1e3
            net.set_conv(fmap)
            d = {"ms_per_image": ms, "proposals_per_s": 300e3 / ms, "t_min_us": fl, "path_floor": floors(std, int(fmap.size), ms * 1e3),
                 "path_floor_frac": floors(std, int(fmap.size), ms * 1e3)["frac"],
                 "queued": {"ms_per_image": ms_q, "proposals_per_s": 300e3 / ms_q, "searches_launched_and_unfetched": dq,
                            "path_floor": floors(std, int(fmap.size), ms_q * 1e3),
                            "note": "the same searches queued ahead on the context's lanes, as `value` is measured"},
                 "kernel_ms_per_image": kms, "kernels_ms": {k: round(v, 4) for k, v in sorted(by.items())},
                 "rows_per_pass": [int(x) for x in list(std.pass_rows)[:int(std.n_passes)]]}
            if out_d is None:
                out_d = dict(d, workload="BASELINE config 4: 800x1200 image (scale %g, conv5_3 %dx%d), K = 7, Tz = 0"
                                         % (scale, fmap.shape[-2], fmap.shape[-1]),
                             regions_per_level=[int(std.level_regions[l]) for l in range(std.n_levels)],
                             unique_per_level=ud, candidates=int(std.n_candidates),
                             form="level by level (what a Tz > 0 search takes)" if not static else "one pass")
            else:
                out_d[form] = d
        return out_d

    res["deep_tree"] = deep(0.75, (("level_loop", False), ("one_pass", True)))
    res["deep_tree_800px_network"] = deep(1.0, (("level_loop", False),))
    # ---- config 3: AZ proposals + Fast R-CNN head (fc6/fc7 4096, 21 classes) + per-class NMS on the shared map -------
    fmap = synth.make_feature_map(4, 512, 38, 63)
    net.set_conv(fmap)
    p = ffi.AzContext.make_params(H_IM, W_IM, 1.0, 0.0, static_tree=False)
    Yp = net.propose(p)
    det = HipDetNet(synth.make_det_head(seed=99), net)

    def det_call():
        return det.detect(Yp, 1.0, (H_IM, W_IM), 1. / 16., 10000, 1e-14)
    sc, bx = det_call()
    # apply_nms's input (test.py:730-770): per class the boxes above the score threshold, <= 100 per class
    groups = []
    for c in range(1, sc.shape[1]):
        o = np.argsort(-sc[:, c])[:100]
        groups.append(np.hstack([bx[o, 4 * c:4 * c + 4], sc[o, c:c + 1]]).astype(np.float32))
    ms_prop = wall(lambda: net.propose(p), n_img)
    ms_det = wall(det_call, n_img)
    ms_nms = wall(lambda: ctx.nms_batched(groups, 0.3), n_img)
    k_det, by_det = kernel_ms(det_call, 5)
    k_nms, _ = kernel_ms(lambda: ctx.nms_batched(groups, 0.3), 5)
    res["shared_detection"] = {
        "workload": "BASELINE config 3: az_propose (level loop) + az_detect on its 300 proposals (fc6/fc7 4096, 21 classes) + "
                    "az_nms_batched (20 classes x 100 boxes, thresh 0.3), 600x1000",
        "az_propose_ms": ms_prop, "az_detect_ms": ms_det, "nms_batched_ms": ms_nms,
        "images_per_s": 1e3 / (ms_prop + ms_det + ms_nms),
        "az_detect_kernel_ms": k_det, "nms_batched_kernel_ms": k_nms,
        "az_detect_kernels_ms": {k: round(v, 4) for k, v in sorted(by_det.items())}}
    # ---- az_nms (lib/utils/nms.pyx), uniform boxes, distinct scores, thresh 0.5 ------------------------------------
    rng = np.random.RandomState(0)
    nms = {}
    for n in (100, 300, 2000, 8129):
        x1 = rng.uniform(0, 900, n)
        y1 = rng.uniform(0, 500, n)
        dets = np.stack([x1, y1, x1 + rng.uniform(10, 210, n), y1 + rng.uniform(10, 210, n),
                         rng.permutation(n) / float(n)], 1).astype(np.float32)
        w = wall(lambda: ctx.nms(dets, 0.5), 20)
        km, _ = kernel_ms(lambda: ctx.nms(dets, 0.5), 10)
        # HBM-side bytes of the kernels: 20 B/box in, the N^2/8-byte suppression matrix written and read once
        alg = 20.0 * n + (0 if n <= 256 else 2.0 * n * ((n + 63) // 64) * 8)
        nms[str(n)] = {"wall_ms_incl_copies": w, "kernel_ms": km, "kept": int(len(ctx.nms(dets, 0.5))),
                       "algorithmic_bytes": alg, "gb_per_s": alg / (km * 1e-3) / 1e9 if km > 0 else None}
    res["nms"] = nms
    return res
