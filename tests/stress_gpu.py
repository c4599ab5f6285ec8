#!/usr/bin/env python3
"""Randomised cross-check of the search paths on the GPU: for random image shapes / Tz / BATCH_SIZE /
MAX_SIZE / proposal counts the default path (speculative levels 1-3, fused geometry, counting top-k) must
equal, bit for bit, the plain path (level by level, multi-launch geometry, radix select), and every fourth
case is also compared with the oracle's loop driven by the HIP head.
`run_case(net, case)` is what tests/test_gpu_stress.py parametrises (the driver runs it under pytest -m gpu);
as a script: stress_gpu.py [n_cases] [first_case] for longer soaks."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "az-net_amd", "lib"))
sys.path.insert(0, os.path.join(HERE, ".."))


FULL = bool(int(os.environ.get("AZ_STRESS_FULL", "0")))     # the full-size head (25088 -> 4096 -> ...): many-row GEMM, big passes
LANES = int(os.environ.get("AZ_STRESS_LANES", "1"))          # 2: az_set_lanes(2) -- the queued searches of a case overlap on the GPU


def make_net():
    from aznet_hip import synth
    from aznet_hip.net import HipAZNet
    if FULL:
        net = HipAZNet(synth.make_head(seed=1234, **synth.FULL_DIMS), name="stress_full", max_regions=4096)
    else:
        net = HipAZNet(synth.make_head(seed=77, **synth.SMALL_DIMS), name="stress")
    net.ctx.set_lanes(LANES)
    return net


def run_case(net, case):
    """One random configuration, seeded by `case`.  Returns (ok, description, reason)."""
    from aznet_hip import ffi, synth
    from oracle import az_oracle as orc
    rng = np.random.RandomState(1000003 * (case + 1) % (2 ** 31 - 1))
    H = int(rng.randint(40, 900)); W = int(rng.randint(40, 1300))
    max_size = int(rng.choice([1000, 800, 600]))
    scale = 600.0 / min(H, W)
    if np.round(scale * max(H, W)) > max_size:
        scale = float(max_size) / max(H, W)
    fh = synth.conv_out_size(int(round(H * scale))); fw = synth.conv_out_size(int(round(W * scale)))
    if orc.num_levels(H, W) - 1 < 1:
        return True, "case %d skipped (%dx%d: no level)" % (case, H, W), ""
    fmap = synth.make_feature_map(100 + case, (synth.FULL_DIMS if FULL else synth.SMALL_DIMS)["C"], fh, fw)
    net.set_conv(fmap)
    batch = int(rng.choice([10000, 10000, 1000, 100, 37]))
    nprop = int(rng.choice([300, 300, 2000, 50]))
    fixed = bool(rng.rand() < 0.85)
    dedup = 0.0 if rng.rand() < 0.1 else 1. / 16.           # cfg.DEDUP_BOXES <= 0: no feature-space dedup
    # Tz from the zoom distribution of this image
    net.propose(ffi.AzContext.make_params(H, W, scale, 0.0, num_proposals=300, batch_size=batch, tune=True))
    z = net.ctx.last_anchors()[1].astype(np.float64)
    Tz = float(rng.choice([0.0, np.quantile(z, rng.uniform(0.2, 0.9)), z[rng.randint(z.size)], 1.5]))
    Tc = float(np.quantile(z, 0.5))
    kw = dict(num_proposals=nprop, batch_size=batch, fixed_num=fixed, Tc=Tc, dedup=dedup)
    desc = "case %d: H=%d W=%d scale=%.4f Tz=%r batch=%d nprop=%d fixed=%s dedup=%g" % (
        case, H, W, scale, Tz, batch, nprop, fixed, dedup)
    a = net.propose(ffi.AzContext.make_params(H, W, scale, Tz, **kw), want_scores=True, want_stats=True)
    Ya, Sa = net.ctx.last_candidates()
    b = net.propose(ffi.AzContext.make_params(H, W, scale, Tz, speculate=False, fused=False, radix_select=True, **kw),
                    want_scores=True, want_stats=True)
    Yb, Sb = net.ctx.last_candidates()
    ok = (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(Ya, Yb) and
          np.array_equal(Sa, Sb) and a[2].num_eval == b[2].num_eval and a[2].depth == b[2].depth)
    if not ok:
        return False, desc, "fused-vs-plain: Y %s S %s Yall %s Sall %s eval %d/%d depth %d/%d" % (
            np.array_equal(a[0], b[0]), np.array_equal(a[1], b[1]), np.array_equal(Ya, Yb), np.array_equal(Sa, Sb),
            a[2].num_eval, b[2].num_eval, a[2].depth, b[2].depth)
    # pair speculation forced at every eligible level (round 3), alone and as the second of two queued searches
    pp = ffi.AzContext.make_params(H, W, scale, Tz, pair_spec=True, **kw)
    c = net.propose(pp, want_scores=True, want_stats=True)
    Yc, Sc = net.ctx.last_candidates()
    ok = (np.array_equal(c[0], b[0]) and np.array_equal(c[1], b[1]) and np.array_equal(Yc, Yb) and
          np.array_equal(Sc, Sb) and c[2].num_eval == b[2].num_eval and c[2].depth == b[2].depth and
          list(c[2].level_unique) == list(b[2].level_unique) and list(c[2].level_zoomed) == list(b[2].level_zoomed))
    if not ok:
        return False, desc, "pair-vs-plain: Y %s S %s Yall %s Sall %s eval %d/%d passes %s" % (
            np.array_equal(c[0], b[0]), np.array_equal(c[1], b[1]), np.array_equal(Yc, Yb), np.array_equal(Sc, Sb),
            c[2].num_eval, b[2].num_eval, list(c[2].pass_rows[:c[2].n_passes]))
    # whole-tree speculation forced (one head pass over the image shape's full tree, outputs by window lookup; a pruned tree
    # that needs a window the pass lacks is repeated level by level): same bits either way
    if fixed:
        pf = ffi.AzContext.make_params(H, W, scale, Tz, full_spec=True, **kw)
        e = net.propose(pf, want_scores=True, want_stats=True)
        Ye, Se = net.ctx.last_candidates()
        ok = (np.array_equal(e[0], b[0]) and np.array_equal(e[1], b[1]) and np.array_equal(Ye, Yb) and
              np.array_equal(Se, Sb) and e[2].num_eval == b[2].num_eval and e[2].depth == b[2].depth and
              list(e[2].level_unique) == list(b[2].level_unique) and list(e[2].level_zoomed) == list(b[2].level_zoomed))
        if not ok:
            return False, desc, "whole-tree-vs-plain: Y %s S %s Yall %s Sall %s eval %d/%d passes %s" % (
                np.array_equal(e[0], b[0]), np.array_equal(e[1], b[1]), np.array_equal(Ye, Yb), np.array_equal(Se, Sb),
                e[2].num_eval, b[2].num_eval, list(e[2].pass_rows[:e[2].n_passes]))
        desc += " wt-passes %d" % e[2].n_passes
        # ... and over the closure rows (every region any pruning can produce): same bits, and NEVER a search run twice
        pcl = ffi.AzContext.make_params(H, W, scale, Tz, full_spec="closure", **kw)
        f = net.propose(pcl, want_scores=True, want_stats=True)
        Yf, Sf = net.ctx.last_candidates()
        ok = (np.array_equal(f[0], b[0]) and np.array_equal(f[1], b[1]) and np.array_equal(Yf, Yb) and
              np.array_equal(Sf, Sb) and f[2].num_eval == b[2].num_eval and f[2].depth == b[2].depth and
              list(f[2].level_unique) == list(b[2].level_unique) and list(f[2].level_zoomed) == list(b[2].level_zoomed))
        if not ok or (f[2].search_form == 3 and f[2].n_reruns != 0):
            return False, desc, "closure-vs-plain: Y %s S %s Yall %s Sall %s eval %d/%d passes %s form %d reruns %d" % (
                np.array_equal(f[0], b[0]), np.array_equal(f[1], b[1]), np.array_equal(Yf, Yb), np.array_equal(Sf, Sb),
                f[2].num_eval, b[2].num_eval, list(f[2].pass_rows[:f[2].n_passes]), f[2].search_form, f[2].n_reruns)
        desc += " closure-form %d rows %d" % (f[2].search_form, f[2].pass_rows[0])
    if fixed:
        # queued searches in all forms -- two deep on one lane, four deep on two lanes (AZ_STRESS_LANES=2: they then overlap
        # on the GPU, and a rerun inside fetch happens while the other lane works)
        forms = [ffi.AzContext.make_params(H, W, scale, Tz, **kw), pp, pf, pcl]
        depth = 2 * LANES
        got, q = [], 0
        for j in range(len(forms) + depth - 1):
            if j < len(forms):
                net.ctx.propose_launch(forms[j])
                q += 1
            if q == depth or j >= len(forms):
                got.append(net.ctx.propose_fetch(want_scores=True))
                q -= 1
        if not all(np.array_equal(g[0], b[0]) and np.array_equal(g[1], b[1]) for g in got) or len(got) != len(forms):
            return False, desc, "queued searches differ from the plain one"
    if fixed:
        # a lockstep batch (az_batch_launch) of this image and up to four more maps of its shape, with the case's settings:
        # every image as its plain search gives it (a BATCH_SIZE that chunks a level's dedup, a shape with two levels, an
        # overflowing table: those batches end up searched image by image or rerun, same results)
        import torch
        nb = int(rng.randint(1, 6))
        C = (synth.FULL_DIMS if FULL else synth.SMALL_DIMS)["C"]
        # (most of the extra images have a shape of their own -- a few pixels off, or half / double the size, so that the
        #  numbers of levels differ as well: az_batch_launch_shapes)
        geo = [(H, W, scale, fh, fw)]
        for j in range(nb - 1):
            H2, W2 = H, W
            u = rng.rand()
            if u < 0.45:
                H2 = max(40, H + int(rng.randint(-24, 25))); W2 = max(40, W + int(rng.randint(-24, 25)))
            elif u < 0.6:
                H2 = max(40, H // 2); W2 = max(40, W // 2)
            elif u < 0.7:
                H2 = min(900, H * 2); W2 = min(1300, W * 2)
            s2 = 600.0 / min(H2, W2)
            if np.round(s2 * max(H2, W2)) > max_size:
                s2 = float(max_size) / max(H2, W2)
            if orc.num_levels(H2, W2) - 1 < 1:
                H2, W2, s2 = H, W, scale
            geo.append((H2, W2, s2, synth.conv_out_size(int(round(H2 * s2))), synth.conv_out_size(int(round(W2 * s2)))))
        maps = [fmap] + [synth.make_feature_map(50000 + 7 * case + j, C, geo[j + 1][3], geo[j + 1][4]) for j in range(nb - 1)]
        want = [b]
        for m, (H2, W2, s2, _, _) in zip(maps[1:], geo[1:]):
            net.set_conv(m)
            want.append(net.propose(ffi.AzContext.make_params(H2, W2, s2, Tz, speculate=False, fused=False, radix_select=True, **kw),
                                    want_scores=True, want_stats=True))
        order = [int(x) for x in rng.permutation(nb)]
        net.ctx.batch_launch([ffi.AzContext.make_params(geo[j][0], geo[j][1], geo[j][2], Tz, **kw) for j in order],
                             [torch.from_numpy(maps[j]).cuda() for j in order])
        for i, j in enumerate(order):
            g = net.ctx.batch_fetch(i, want_scores=True, want_stats=True)
            if not (np.array_equal(g[0], want[j][0]) and np.array_equal(g[1], want[j][1]) and g[2].num_eval == want[j][2].num_eval and
                    list(g[2].level_unique) == list(want[j][2].level_unique) and list(g[2].level_zoomed) == list(want[j][2].level_zoomed)):
                return False, desc, "image %d of a lockstep batch of %d differs from its plain search (form %d reruns %d)" % (
                    i, nb, g[2].search_form, g[2].n_reruns)
        net.set_conv(fmap)
        desc += " batch %d" % nb
    desc += " levels %d eval %d cand %d" % (a[2].n_levels, a[2].num_eval, Ya.shape[0])
    if case % (16 if FULL else 4):
        return True, desc, ""

    class Inj(object):
        name = "inj"; blobs = net.blobs

        def forward(self, blobs=None, **k2):
            k2.pop("data", None); k2["conv5_3"] = fmap
            return net.forward(blobs=blobs, **k2)
    inj = Inj()
    cfg = orc.OracleCfg(Tz=Tz, BATCH_SIZE=batch, NUM_PROPOSALS=nprop, FIXED_PROPOSAL_NUM=fixed, Tc=Tc,
                        DEDUP_BOXES=dedup)
    Yref, tr = orc.im_propose({"full": inj, "fc": inj}, (H, W), scale, cfg, return_trace=True)
    # boxes: 1 ulp of the f32 exp (NumPy's SIMD expf vs the device's) times half a predicted width of up to ~2500 px
    ok = (a[2].num_eval == tr["num_eval"] and Ya.shape == tr["Y_all"].shape and
          np.array_equal(Sa.astype(np.float64), tr["aScores"]) and
          np.allclose(Ya, tr["Y_all"], rtol=1e-6, atol=3e-4) and a[0].shape == Yref.shape)
    if ok:
        return True, desc + " (+oracle loop)", ""
    same = Ya.shape == tr["Y_all"].shape
    dS = np.abs(Sa.astype(np.float64) - tr["aScores"]).max() if same else -1
    dY = np.abs(Ya - tr["Y_all"]).max() if same else -1
    why = "vs oracle loop: max|dS| %.3g max|dY| %.3g Yref %s top %s; eval %d/%d shapes %s/%s levels gpu %s oracle %s uniq gpu %s oracle %s" % (
        dS, dY, Yref.shape, a[0].shape, a[2].num_eval, tr["num_eval"], Ya.shape, tr["Y_all"].shape,
        [int(a[2].level_regions[l]) for l in range(a[2].n_levels)], [lv["B"].shape[0] for lv in tr["levels"]],
        [int(a[2].level_unique[l]) for l in range(a[2].n_levels)],
        [sum(f["U"] for f in lv["fwd"]) for lv in tr["levels"]])
    return False, desc, why


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    net = make_net()
    bad = 0
    for case in range(first, first + n_cases):
        try:
            ok, desc, why = run_case(net, case)
        except Exception as e:                       # a tree that outgrows the context's limits is not a mismatch
            if getattr(e, "code", None) == -3:
                print("skipped case %d: %s" % (case, str(e)[:80]))
                net = make_net()                     # (a failed search leaves nothing queued, but start clean)
                continue
            raise
        if not ok:
            bad += 1
            print("MISMATCH %s | %s" % (desc, why))
        elif case % 20 == 0:
            print("ok %s" % desc)
    print("stress: %d cases, %d mismatches" % (n_cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
