"""A batch of images of one shape searched in lockstep (az_batch_launch; az-net_amd/csrc/az_batch.hip, az_search.hip:
batch_launch_impl) -- the images of consecutive iterations of the reference's dataset loop (lib/detect/test.py:508-513), each
with its own tree, every level's rois of all of them in ONE head pass.  Whatever shares a pass with an image, its result is
what the plain level loop (and the CPU oracle) give for that image alone: boxes, scores, every counter, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    import torch
    from aznet_hip import ffi, synth
    from aznet_hip.net import HipAZNet
    from oracle import az_oracle as orc
    return torch, ffi, synth, HipAZNet, orc


def _plain(ffi, H, W, scale, Tz, **kw):
    return ffi.AzContext.make_params(H, W, scale, Tz, speculate=False, fused=False, fused_levels=False, static_tree=False,
                                     pair_spec=False, full_spec=False, early_end=False, **kw)


def _cl(torch, fmap):
    return torch.from_numpy(np.ascontiguousarray(fmap)).cuda().contiguous(memory_format=torch.channels_last)


def _same(st, sr):
    assert st.n_levels == sr.n_levels and st.num_eval == sr.num_eval and st.depth == sr.depth
    assert st.n_candidates == sr.n_candidates and st.n_proposals == sr.n_proposals
    for l in range(sr.n_levels):
        assert st.level_regions[l] == sr.level_regions[l] and st.level_unique[l] == sr.level_unique[l]
        assert st.level_zoomed[l] == sr.level_zoomed[l]


def _covered(st):
    cover = 0
    for i in range(int(st.n_passes)):
        assert cover & int(st.pass_levels[i]) == 0
        cover |= int(st.pass_levels[i])
    for l in range(st.n_levels):
        if st.level_unique[l] > 0:
            assert (cover >> l) & 1


def _object_set(synth, n, H=600, W=1000, scale=1.0):
    fh, fw = synth.conv_out_size(int(round(H * scale))), synth.conv_out_size(int(round(W * scale)))
    return [synth.make_object_map(j, synth.SMALL_DIMS["C"], fh, fw) for j in range(n)]


def _reference(ffi, ref, H, W, scale, Tz, fmaps, **kw):
    out = []
    for f in fmaps:
        ref.set_conv(f)
        out.append(ref.propose(_plain(ffi, H, W, scale, Tz, **kw), want_scores=True, want_stats=True))
    return out


TZ_OBJ = 0.6226829886436462        # (tests/golden/g14_stream.npz: the reference's tuned threshold over the object images)


@pytest.mark.parametrize("n", [1, 2, 5, 8, 16, 32])
def test_batch_equals_every_image_alone(mods, n):
    torch, ffi, synth, HipAZNet, orc = mods
    head = synth.make_object_head(seed=77, **synth.SMALL_DIMS)
    H, W, sc = 600, 1000, 1.0
    fmaps = _object_set(synth, 32)[:n]
    ref = HipAZNet(head, name="batch_ref")
    want = _reference(ffi, ref, H, W, sc, TZ_OBJ, fmaps)
    assert len({tuple(int(s.level_regions[l]) for l in range(s.n_levels)) for _, _, s in want}) >= min(n, 4) or n < 4
    net = HipAZNet(head, name="batch")
    prm = ffi.AzContext.make_params(H, W, sc, TZ_OBJ)
    tm = [_cl(torch, f) for f in fmaps]
    for rep in range(3):                       # (the slots are used again: nothing of the previous batch may leak)
        net.ctx.batch_launch(prm, tm if rep < 2 else tm[::-1], producer_done=True)
        order = list(range(n)) if rep < 2 else list(range(n))[::-1]
        for i in range(n):
            Y, S, st = net.ctx.batch_fetch(i, want_scores=True, want_stats=True)
            Yr, Sr, sr = want[order[i]]
            assert np.array_equal(Y, Yr) and np.array_equal(S, Sr), (rep, i)
            _same(st, sr)
            assert st.search_form == 5 and st.n_reruns == 0
            assert st.n_passes == sum(1 for l in range(2, st.n_levels) if st.level_unique[l] > 0) + 1
            assert st.pass_rows[0] == 9 and st.pass_levels[0] == 3
            _covered(st)


def test_fetch_all_equals_fetch_one_by_one(mods):
    torch, ffi, synth, HipAZNet, orc = mods
    head = synth.make_object_head(seed=77, **synth.SMALL_DIMS)
    H, W, sc = 600, 1000, 1.0
    tm = [_cl(torch, f) for f in _object_set(synth, 9)]
    net = HipAZNet(head, name="fetch_all")
    prm = ffi.AzContext.make_params(H, W, sc, TZ_OBJ)
    net.ctx.batch_launch(prm, tm, producer_done=True)
    one = [net.ctx.batch_fetch(i, want_scores=True, want_stats=True) for i in range(9)]
    net.ctx.batch_launch(prm, tm, producer_done=True)
    net.ctx.batch_launch(prm, tm[:4], producer_done=True)
    every = net.ctx.batch_fetch_all(want_scores=True, want_stats=True)
    part = net.ctx.batch_fetch_all(want_scores=True)
    assert len(every) == 9 and len(part) == 4
    for (Y, S, st), (Y1, S1, st1) in zip(every, one):
        assert np.array_equal(Y, Y1) and np.array_equal(S, S1)
        _same(st, st1)
        assert st.search_form == 5
    for (Y, S), (Y1, S1, _) in zip(part, one):
        assert np.array_equal(Y, Y1) and np.array_equal(S, S1)
    with pytest.raises(ffi.AzError):
        net.ctx.batch_fetch_all()


def test_batch_against_the_cpu_oracle(mods):
    torch, ffi, synth, HipAZNet, orc = mods
    head = synth.make_object_head(seed=77, **synth.SMALL_DIMS)
    H, W, sc = 600, 1000, 1.0
    fmaps = _object_set(synth, 6)
    net = HipAZNet(head, name="batch_orc")
    net.ctx.batch_launch(ffi.AzContext.make_params(H, W, sc, TZ_OBJ), [_cl(torch, f) for f in fmaps], producer_done=True)
    for i, f in enumerate(fmaps):
        Y, S, st = net.ctx.batch_fetch(i, want_scores=True, want_stats=True)
        onet = orc.OracleNet(head, feat_fn=lambda d, f=f: f)
        Yo, tr = orc.im_propose({"full": onet, "fc": onet}, (H, W), sc, orc.OracleCfg(Tz=TZ_OBJ), return_trace=True)
        assert [int(st.level_regions[l]) for l in range(len(tr["levels"]))] == [lv["B"].shape[0] for lv in tr["levels"]]
        assert st.num_eval == tr["num_eval"] and st.depth == tr["depth"] and st.n_candidates == tr["Y_all"].shape[0]
        assert Y.shape == Yo.shape
        hit = [np.abs(Y - r).max(axis=1).min() <= 1e-3 for r in Yo]
        assert np.mean(hit) >= 0.97


@pytest.mark.parametrize("shape", [(375, 500, 1.6), (480, 640, 1.25), (333, 500, 600.0 / 333), (600, 1000, 1.0)])
@pytest.mark.parametrize("tz", [0.0, 0.35, 0.9])
def test_batch_shapes_and_thresholds(mods, shape, tz):
    """Random weights (the trees are whatever the threshold makes of them: full at 0, the root's children only at 0.9)."""
    torch, ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    H, W, sc = shape
    fh, fw = synth.conv_out_size(int(round(H * sc))), synth.conv_out_size(int(round(W * sc)))
    fmaps = [synth.make_scene_map(j, synth.SMALL_DIMS["C"], fh, fw) for j in range(5)]
    ref = HipAZNet(head, name="bs_ref")
    want = _reference(ffi, ref, H, W, sc, tz, fmaps)
    net = HipAZNet(head, name="bs")
    net.ctx.batch_launch(ffi.AzContext.make_params(H, W, sc, tz, static_tree=False), [_cl(torch, f) for f in fmaps], producer_done=True)
    for i in range(5):
        Y, S, st = net.ctx.batch_fetch(i, want_scores=True, want_stats=True)
        assert np.array_equal(Y, want[i][0]) and np.array_equal(S, want[i][1]), i
        _same(st, want[i][2])


def test_a_level_that_outgrows_the_head_buffers(mods):
    """max_regions 1024 and five full trees of 517 unique rois at the last level: the pass does not fit, every image of the
    batch is run again on its own by batch_fetch -- same results, n_reruns 1.  The next batch on the lane goes by the rows
    the last one had: it is enqueued in parts that fit (here: one image each), nothing is run twice."""
    torch, ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    H, W, sc = 600, 1000, 1.0
    fmaps = [synth.make_scene_map(j, synth.SMALL_DIMS["C"], 38, 63) for j in range(5)]
    ref = HipAZNet(head, name="ovf_ref", max_regions=1024)
    want = _reference(ffi, ref, H, W, sc, 0.0, fmaps)
    assert int(want[0][2].level_regions[4]) == 564 and int(want[0][2].level_unique[4]) == 517
    net = HipAZNet(head, name="ovf", max_regions=1024)
    for rep in range(3):
        net.ctx.batch_launch(ffi.AzContext.make_params(H, W, sc, 0.0, static_tree=False), [_cl(torch, f) for f in fmaps], producer_done=True)
        for i in range(5):
            Y, S, st = net.ctx.batch_fetch(i, want_scores=True, want_stats=True)
            assert np.array_equal(Y, want[i][0]) and np.array_equal(S, want[i][1]), i
            _same(st, want[i][2])
            if rep == 0:
                assert st.n_reruns == 1 and st.search_form != 5
            else:
                assert st.n_reruns == 0 and st.search_form == 5


def test_shapes_the_lockstep_form_does_not_take(mods):
    """A 40x60 image has two levels: its batch is searched image by image, same results."""
    torch, ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    H, W, sc = 40, 60, 15.0
    fh, fw = synth.conv_out_size(int(round(H * sc))), synth.conv_out_size(int(round(W * sc)))
    fmaps = [synth.make_scene_map(j, synth.SMALL_DIMS["C"], fh, fw) for j in range(3)]
    ref = HipAZNet(head, name="small_ref")
    want = _reference(ffi, ref, H, W, sc, 0.3, fmaps)
    assert want[0][2].n_levels < 3
    net = HipAZNet(head, name="small")
    net.ctx.batch_launch(ffi.AzContext.make_params(H, W, sc, 0.3), [_cl(torch, f) for f in fmaps], producer_done=True)
    for i in range(3):
        Y, S, st = net.ctx.batch_fetch(i, want_scores=True, want_stats=True)
        assert np.array_equal(Y, want[i][0]) and np.array_equal(S, want[i][1])
        assert st.search_form != 5


@pytest.mark.parametrize("lanes", [1, 2])
def test_batches_in_flight_and_single_searches_in_between(mods, lanes):
    """Two batches per lane may be launched before the first is fetched (a lane's two run one after the other on its stream,
    those of two lanes interleave on the GPU); results come back oldest first."""
    torch, ffi, synth, HipAZNet, orc = mods
    head = synth.make_object_head(seed=77, **synth.SMALL_DIMS)
    H, W, sc = 600, 1000, 1.0
    fmaps = _object_set(synth, 24)
    ref = HipAZNet(head, name="two_ref")
    want = _reference(ffi, ref, H, W, sc, TZ_OBJ, fmaps)
    net = HipAZNet(head, name="two")
    net.ctx.set_lanes(lanes)
    inflight = 2 * lanes
    prm = ffi.AzContext.make_params(H, W, sc, TZ_OBJ)
    tm = [_cl(torch, f) for f in fmaps]
    with pytest.raises(ffi.AzError):
        net.ctx.batch_fetch(0)                                         # nothing launched
    groups = [list(range(0, 8)), list(range(8, 16)), list(range(16, 24)), [3, 4, 5], [23], [1, 0], list(range(4, 20)), [7]]
    launched = 0
    for gi in range(len(groups)):
        while launched < min(len(groups), gi + inflight):
            net.ctx.batch_launch(prm, [tm[j] for j in groups[launched]], producer_done=True)
            launched += 1
        if gi == 0:
            with pytest.raises(ffi.AzError):
                net.ctx.batch_launch(prm, [tm[0]], producer_done=True)     # every lane holds two batches
            with pytest.raises(ffi.AzError):
                net.ctx._chk(net.ctx.L.az_batch_fetch(net.ctx.h, 1, None, None, 0, None, None))
        for i, j in enumerate(groups[gi]):
            Y, S, st = net.ctx.batch_fetch(i, want_scores=True, want_stats=True)
            assert np.array_equal(Y, want[j][0]) and np.array_equal(S, want[j][1]), (gi, i)
            _same(st, want[j][2])
        if gi == 1:
            # a search launched the usual way between two batches, on the same context
            net.ctx.propose_launch(prm, fmap=tm[7], producer_done=True)
            Y, S = net.ctx.propose_fetch(want_scores=True)
            assert np.array_equal(Y, want[7][0]) and np.array_equal(S, want[7][1])


def test_batch_with_the_full_head(mods, gemm_mode):
    """Launch sizes of the head (512 channels, int6 25088 -> 4096): 8 object images in lockstep against each alone -- on the
    fp32 MFMA and with int6 as three bf16 terms per operand (no per-map scale: the images share passes there as well)."""
    torch, ffi, synth, HipAZNet, orc = mods
    head = synth.make_object_head(seed=1234, **synth.FULL_DIMS)
    H, W, sc = 600, 1000, 1.0
    fmaps = [synth.make_object_map(j, 512, 38, 63) for j in range(8)]
    net = HipAZNet(head, name="full_batch", max_regions=4096, gemm_mode=gemm_mode)
    ref = net
    # a threshold from the set itself, as the tuner gives it (ANCHORS_PER_IMG = 20)
    net.ctx.tune_begin(8 * 2 * net.ctx.max_regions)
    for f in fmaps:
        net.set_conv(f)
        net.propose(ffi.AzContext.make_params(H, W, sc, 0.0, tune=True))
    tz, _ = net.ctx.tune_kth_largest(8 * 20)
    net.ctx.tune_end()
    tz = float(tz) - 1e-4
    want = _reference(ffi, ref, H, W, sc, tz, fmaps)
    trees = {tuple(int(s.level_regions[l]) for l in range(s.n_levels)) for _, _, s in want}
    assert len(trees) >= 4
    net.ctx.batch_launch(ffi.AzContext.make_params(H, W, sc, tz), [_cl(torch, f) for f in fmaps], producer_done=True)
    for i in range(8):
        Y, S, st = net.ctx.batch_fetch(i, want_scores=True, want_stats=True)
        assert np.array_equal(Y, want[i][0]) and np.array_equal(S, want[i][1]), i
        _same(st, want[i][2])
        assert st.search_form == 5 and st.n_reruns == 0


def test_two_fp16_terms_keep_their_images_apart(mods):
    """az_set_gemm_mode 2 scales pool5 per map: a batch there is searched image by image, same results."""
    torch, ffi, synth, HipAZNet, orc = mods
    head = synth.make_object_head(seed=77, **synth.SMALL_DIMS)
    H, W, sc = 600, 1000, 1.0
    fmaps = _object_set(synth, 4)
    net = HipAZNet(head, name="m2", gemm_mode=2)
    want = _reference(ffi, net, H, W, sc, TZ_OBJ, fmaps)
    net.ctx.batch_launch(ffi.AzContext.make_params(H, W, sc, TZ_OBJ), [_cl(torch, f) for f in fmaps], producer_done=True)
    for i in range(4):
        Y, S, st = net.ctx.batch_fetch(i, want_scores=True, want_stats=True)
        assert np.array_equal(Y, want[i][0]) and np.array_equal(S, want[i][1])
        assert st.search_form != 5


def test_batch_entry_points_refuse_what_they_cannot_do(mods):
    """Argument and state errors come back as AZ_ERR_* with a message, nothing is left half-launched."""
    import ctypes
    torch, ffi, synth, HipAZNet, orc = mods
    head = synth.make_object_head(seed=77, **synth.SMALL_DIMS)
    H, W, sc = 600, 1000, 1.0
    tm = [_cl(torch, f) for f in _object_set(synth, 3)]
    net = HipAZNet(head, name="errs")
    ctx, L = net.ctx, net.ctx.L
    prm = ffi.AzContext.make_params(H, W, sc, TZ_OBJ)
    ptrs = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in tm])
    C = synth.SMALL_DIMS["C"]

    def rc(*a):
        return L.az_batch_launch(ctx.h, *a)
    assert rc(0, ctypes.byref(prm), ptrs, C, 38, 63) == ffi.AZ_ERR_INVALID                      # no image
    assert rc(ffi.AZ_BATCH_MAX + 1, ctypes.byref(prm), ptrs, C, 38, 63) == ffi.AZ_ERR_INVALID   # too many
    assert rc(3, ctypes.byref(prm), ptrs, C + 4, 38, 63) == ffi.AZ_ERR_INVALID                  # another head's channel count
    assert rc(3, None, ptrs, C, 38, 63) == ffi.AZ_ERR_INVALID
    bad = (ctypes.c_void_p * 3)(tm[0].data_ptr(), None, tm[2].data_ptr())
    assert rc(3, ctypes.byref(prm), bad, C, 38, 63) == ffi.AZ_ERR_INVALID                       # a null map
    var = ffi.AzContext.make_params(H, W, sc, TZ_OBJ, fixed_num=False)
    assert rc(3, ctypes.byref(var), ptrs, C, 38, 63) == ffi.AZ_ERR_INVALID                      # data-dependent proposal count
    tune = ffi.AzContext.make_params(H, W, sc, 0.0, tune=True)
    assert rc(3, ctypes.byref(tune), ptrs, C, 38, 63) == ffi.AZ_ERR_INVALID                     # the tuner's variant
    with pytest.raises(ffi.AzError):
        ctx.batch_fetch(0)                                                                      # nothing was launched by any of these
    # a good batch; fetching out of order or twice is refused, the batch itself stays whole
    want = _reference(ffi, HipAZNet(head, name="errs_ref"), H, W, sc, TZ_OBJ, _object_set(synth, 3))
    ctx.batch_launch(prm, tm, producer_done=True)
    boxes = np.empty((300, 4)); n = ctypes.c_int(0)
    bp = boxes.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert L.az_batch_fetch(ctx.h, 1, bp, None, 300, ctypes.byref(n), None) == ffi.AZ_ERR_INVALID
    assert L.az_batch_fetch(ctx.h, 0, bp, None, 10, ctypes.byref(n), None) == ffi.AZ_ERR_CAPACITY   # cap too small: image 0 is gone ...
    with pytest.raises(ffi.AzError):
        ctx.set_lanes(2)                                                                        # ... but the batch is still in flight
    Y1 = ctx.batch_fetch(1)
    Y2 = ctx.batch_fetch(2)
    assert np.array_equal(Y1, want[1][0]) and np.array_equal(Y2, want[2][0])
    with pytest.raises(ffi.AzError):
        ctx.batch_fetch(2)
    ctx.set_lanes(2)
    ctx.set_lanes(1)
    # and the context is as good as new
    ctx.batch_launch(prm, tm, producer_done=True)
    for i, (Y, S) in enumerate(ctx.batch_fetch_all(want_scores=True)):
        assert np.array_equal(Y, want[i][0]) and np.array_equal(S, want[i][1])
    # a launched, unfetched search WITHOUT a fixed proposal count (its counters are still on the device: the batch's pre-pass
    # and passes would overwrite them) refuses the batch, as launch_impl refuses to queue a search behind it; fetched, it
    # has its own result and the batch goes through
    net.set_conv(tm[0])
    ctx.propose_launch(var)
    assert rc(3, ctypes.byref(prm), ptrs, C, 38, 63) == ffi.AZ_ERR_STATE
    Yv = ctx.propose_fetch()
    ref = HipAZNet(head, name="errs_ref2")
    ref.set_conv(tm[0])
    assert np.array_equal(Yv, ref.propose(var))
    ctx.batch_launch(prm, tm, producer_done=True)
    for i, (Y, S) in enumerate(ctx.batch_fetch_all(want_scores=True)):
        assert np.array_equal(Y, want[i][0]) and np.array_equal(S, want[i][1])


def test_slots_searching_on_their_own_share_one_head_set(mods):
    """A batch whose images are searched one by one (here: a level outgrows max_regions, every image is rerun by
    batch_fetch; then a shape the lockstep form does not take) must not allocate a head-buffer set per slot -- at the CLI's
    max_regions and the VGG16 dims that is ~8.5 GB per slot, 32 slots per set, two sets per lane: the slots take the
    context's ONE spare set in turn.  Device memory in use grows by less than two sets' worth over a 6-image batch."""
    torch, ffi, synth, HipAZNet, orc = mods
    dims = dict(C=64, n6=2048, n71=256, n72=64)         # (a head set of 334 MB at 2048 regions: six of them would show)
    head = synth.make_head(seed=77, **dims)
    H, W, sc = 600, 1000, 1.0
    R = 2048
    fmaps = [synth.make_scene_map(j, dims["C"], 38, 63) for j in range(6)]
    ref = HipAZNet(head, name="spare_ref", max_regions=R)
    want = _reference(ffi, ref, H, W, sc, 0.0, fmaps)
    net = HipAZNet(head, name="spare", max_regions=R)
    net.set_conv(fmaps[0])
    net.propose(ffi.AzContext.make_params(H, W, sc, 0.0, static_tree=False))     # (the context's own buffers, plans, calibration)
    K6, n6, n7 = dims["C"] * 49, dims["n6"], dims["n71"] + dims["n72"]
    one_set = 4 * R * (K6 + 16 * n6 + n6 + n7 + 8 * n7)                            # pool5 + slabs + h6 + h7 + int7's slabs
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    net.ctx.batch_launch(ffi.AzContext.make_params(H, W, sc, 0.0, static_tree=False), [_cl(torch, f) for f in fmaps], producer_done=True)
    for i in range(6):
        Y, S, st = net.ctx.batch_fetch(i, want_scores=True, want_stats=True)
        assert np.array_equal(Y, want[i][0]) and np.array_equal(S, want[i][1]), i
        assert st.n_reruns == 1                       # every image ran again on its own
    torch.cuda.synchronize()
    grown = free0 - torch.cuda.mem_get_info()[0]
    # (the six slots' geometry buffers and result blocks are theirs -- a few MB each at these limits; six head sets would be
    #  6 x one_set on top)
    assert grown < 2 * one_set + 6 * (64 << 20), (grown, one_set)
    assert 2 * 6 * one_set > 3 * (2 * one_set + 6 * (64 << 20))  # (the bound separates one set from six at these dims)


@pytest.mark.parametrize("tz", [0.0, 0.3, 0.5])
def test_images_of_several_shapes_in_one_batch(mods, tz):
    """az_batch_launch_shapes: 600x1000, 375x500 (scale 1.6), 500x375, 333x500 and 480x640 images -- five map sizes, five
    pre-passes, five clipping boxes -- share their head passes; every image as its plain search alone gives it."""
    torch, ffi, synth, HipAZNet, orc = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    shapes = [(600, 1000), (375, 500), (500, 375), (333, 500), (480, 640), (375, 500), (600, 1000), (500, 375), (375, 500)]
    items = []
    for j, (H, W) in enumerate(shapes):
        sc = 600.0 / min(H, W)
        if round(sc * max(H, W)) > 1000:
            sc = 1000.0 / max(H, W)
        fh, fw = synth.conv_out_size(int(round(H * sc))), synth.conv_out_size(int(round(W * sc)))
        items.append((H, W, sc, synth.make_scene_map(300 + j, synth.SMALL_DIMS["C"], fh, fw)))
    ref = HipAZNet(head, name="shapes_ref")
    want = []
    for (H, W, sc, f) in items:
        ref.set_conv(f)
        want.append(ref.propose(_plain(ffi, H, W, sc, tz), want_scores=True, want_stats=True))
    net = HipAZNet(head, name="shapes")
    prm = [ffi.AzContext.make_params(H, W, sc, tz, static_tree=False) for (H, W, sc, _) in items]
    tm = [_cl(torch, f) for (_, _, _, f) in items]
    for rep in range(2):
        order = list(range(len(items))) if rep == 0 else list(range(len(items)))[::-1]
        net.ctx.batch_launch([prm[i] for i in order], [tm[i] for i in order], producer_done=True)
        for i, (Y, S, st) in zip(order, net.ctx.batch_fetch_all(want_scores=True, want_stats=True)):
            assert np.array_equal(Y, want[i][0]) and np.array_equal(S, want[i][1]), (rep, i, shapes[i])
            _same(st, want[i][2])
            assert st.search_form == 5 and st.n_reruns == 0
    # images with other numbers of levels in the batch (160x240: four, 90x130: three, against five): an image's last level
    # gets its final selection while the others go on
    extra = []
    for j, (H, W) in enumerate(((160, 240), (90, 130), (1000, 700))):
        sc = 600.0 / min(H, W)
        if round(sc * max(H, W)) > 1000:
            sc = 1000.0 / max(H, W)
        fh, fw = synth.conv_out_size(int(round(H * sc))), synth.conv_out_size(int(round(W * sc)))
        m = synth.make_scene_map(400 + j, synth.SMALL_DIMS["C"], fh, fw)
        ref.set_conv(m)
        extra.append((ffi.AzContext.make_params(H, W, sc, tz, static_tree=False), _cl(torch, m),
                      ref.propose(_plain(ffi, H, W, sc, tz), want_scores=True, want_stats=True)))
    assert sorted({e[2][2].n_levels for e in extra} | {want[0][2].n_levels}) == [3, 4, 5, 6]
    net.ctx.batch_launch([prm[0], extra[0][0], prm[1], extra[1][0], extra[2][0], prm[2]],
                         [tm[0], extra[0][1], tm[1], extra[1][1], extra[2][1], tm[2]], producer_done=True)
    got = net.ctx.batch_fetch_all(want_scores=True, want_stats=True)
    for (Y, S, st), w in zip(got, (want[0], extra[0][2], want[1], extra[1][2], extra[2][2], want[2])):
        assert np.array_equal(Y, w[0]) and np.array_equal(S, w[1])
        _same(st, w[2])
        if max(int(st.level_regions[l]) for l in range(st.n_levels)) <= 1024:
            assert st.search_form == 5 and st.n_reruns == 0
        else:
            # (the 1000x700 image's tree has up to 1447 regions at its sixth level: more than the fused level kernel's tables
            #  hold -- that image is searched again alone, the others are not)
            assert st.n_reruns == 1
    # an image too small for the lockstep form (40x60: two levels): the batch is searched one by one, same results
    H, W, sc = 40, 60, 15.0
    fh, fw = synth.conv_out_size(int(round(H * sc))), synth.conv_out_size(int(round(W * sc)))
    small = synth.make_scene_map(410, synth.SMALL_DIMS["C"], fh, fw)
    ref.set_conv(small)
    w_small = ref.propose(_plain(ffi, H, W, sc, tz), want_scores=True, want_stats=True)
    assert w_small[2].n_levels == 2
    net.ctx.batch_launch([prm[0], ffi.AzContext.make_params(H, W, sc, tz, static_tree=False), prm[1]],
                         [tm[0], _cl(torch, small), tm[1]], producer_done=True)
    for (Y, S, st), w in zip(net.ctx.batch_fetch_all(want_scores=True, want_stats=True), (want[0], w_small, want[1])):
        assert np.array_equal(Y, w[0]) and np.array_equal(S, w[1])
        assert st.search_form != 5


@pytest.mark.parametrize("lockstep", [True, False])
def test_a_batch_staged_into_a_device_buffer(mods, lockstep):
    """az_batch_stage_results_dev: the batch's result records go to rows of a device buffer (the RCCL send buffer of an
    image-sharded run) in one strided copy -- complete when the images have been fetched, equal to what the fetch returns;
    also for a batch that is searched image by image (a 40x60 image: one copy per image there)."""
    torch, ffi, synth, HipAZNet, orc = mods
    head = synth.make_object_head(seed=77, **synth.SMALL_DIMS)
    if lockstep:
        H, W, sc, tz = 600, 1000, 1.0, TZ_OBJ
        fmaps = _object_set(synth, 5)
    else:
        H, W, sc, tz = 40, 60, 15.0, 0.3
        fh, fw = synth.conv_out_size(int(round(H * sc))), synth.conv_out_size(int(round(W * sc)))
        fmaps = [synth.make_scene_map(j, synth.SMALL_DIMS["C"], fh, fw) for j in range(5)]
    net = HipAZNet(head, name="staged")
    k = 100
    prm = ffi.AzContext.make_params(H, W, sc, tz, num_proposals=k)
    nbytes, n_off, b_off, s_off = ffi.AzContext.result_record_layout(k)
    pitch = nbytes + 64
    buf = torch.zeros((7, pitch), dtype=torch.uint8, device="cuda")
    with pytest.raises(ffi.AzError):
        net.ctx.batch_stage_results(buf.data_ptr(), pitch, 7 * pitch)          # no batch in flight
    net.ctx.batch_launch(prm, [_cl(torch, f) for f in fmaps], producer_done=True)
    with pytest.raises(ffi.AzError):
        net.ctx.batch_stage_results(buf.data_ptr(), nbytes - 4, 7 * pitch)     # rows too short
    net.ctx.batch_stage_results(buf[1].data_ptr(), pitch, 6 * pitch)           # rows 1 .. 5
    got = net.ctx.batch_fetch_all(want_scores=True, want_stats=True)
    raw = buf.cpu().numpy()
    assert not raw[0].any() and not raw[6].any()
    for i, (Y, S, st) in enumerate(got):
        assert (st.search_form == 5) == lockstep
        r = raw[1 + i]
        n = int(r[n_off:n_off + 4].view(np.int32)[0])
        assert n == Y.shape[0]
        assert np.array_equal(r[b_off:b_off + n * 32].view(np.float64).reshape(n, 4), Y)
        assert np.array_equal(r[s_off:s_off + n * 4].view(np.float32), S)
        assert not r[nbytes:].any()                                            # (the pitch's padding is not touched)


def test_the_map_cache_of_batch_launch_does_not_keep_maps_alive(mods):
    """batch_launch remembers the outcome of a tensor's layout checks per tensor object; the entry must not hold the tensor
    (nor a view of it): a dataset loop hands over a fresh conv map per image, and 256 pinned 600x1000 maps are 1.25 GB.
    Once a batch is fetched and the caller drops its maps, nothing of them is left."""
    import gc
    import weakref
    torch, ffi, synth, HipAZNet, orc = mods
    head = synth.make_object_head(seed=77, **synth.SMALL_DIMS)
    net = HipAZNet(head, name="cache")
    prm = ffi.AzContext.make_params(600, 1000, 1.0, TZ_OBJ)
    refs = []
    for rep in range(3):
        tm = [_cl(torch, f) for f in _object_set(synth, 4)]           # 4-D channels_last: the layout the batch reads in place
        tm3 = [t[0] for t in tm[:2]]                                   # ... and two of them handed over as 3-D views
        maps = tm3 + tm[2:]
        net.ctx.batch_launch(prm, maps, producer_done=True)
        net.ctx.batch_fetch_all()
        refs += [weakref.ref(t) for t in tm]
        del tm, tm3, maps
    gc.collect()
    assert all(r() is None for r in refs)
    assert len(net.ctx._bmap_cache) <= 64
