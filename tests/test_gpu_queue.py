"""Two searches queued on one context (az_propose_launch twice before az_propose_fetch): the host enqueues the next
image's launch sequence while the GPU works on the current one.  Same stream, so nothing overlaps on the GPU; results
come back oldest first and equal the strictly alternating launch / fetch sequence bit for bit -- also when a search has
to be rerun in another form (NaN zoom score under the one-pass plan, a level that outgrows the fused kernels)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from aznet_hip import ffi, synth
    from aznet_hip.net import HipAZNet
    return ffi, synth, HipAZNet


def _cases(ffi, synth):
    out = []
    for i, (H, W, scale, Tz, static) in enumerate([(600, 1000, 1.0, 0.0, True), (600, 1000, 1.0, 0.3, True),
                                                   (375, 500, 1.6, 0.0, False), (800, 1200, 0.75, 0.0, False),
                                                   (600, 1000, 1.0, 0.0, False), (480, 640, 1.25, 0.2, True)]):
        fh, fw = synth.conv_out_size(int(round(H * scale))), synth.conv_out_size(int(round(W * scale)))
        out.append((ffi.AzContext.make_params(H, W, scale, Tz, static_tree=static),
                    synth.make_feature_map(60 + i, synth.SMALL_DIMS["C"], fh, fw)))
    return out


@pytest.mark.parametrize("nan_zoom", [False, True])
def test_queue_ahead_equals_alternating(mods, nan_zoom):
    import torch
    ffi, synth, HipAZNet = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    if nan_zoom:
        head["bz"] = np.full(1, np.nan, dtype=np.float32)
    cases = _cases(ffi, synth)
    maps = [torch.from_numpy(m).cuda() for _, m in cases]
    ref_net = HipAZNet(head, name="q_ref")
    want = []
    for (p, _), m in zip(cases, maps):
        ref_net.set_conv(m)
        want.append(ref_net.propose(p, want_scores=True, want_stats=True))
    net = HipAZNet(head, name="q_ahead")            # (meets every shape for the first time while queued)
    seq = list(range(len(cases))) * 2
    got = []
    net.ctx.propose_launch(cases[seq[0]][0], fmap=maps[seq[0]])
    for j, i in enumerate(seq):
        if j + 1 < len(seq):
            net.ctx.propose_launch(cases[seq[j + 1]][0], fmap=maps[seq[j + 1]])
        got.append(net.ctx.propose_fetch(want_scores=True, want_stats=True))
    for j, i in enumerate(seq):
        Y, S, st = got[j]
        Yw, Sw, stw = want[i]
        assert np.array_equal(Y, Yw) and np.array_equal(S, Sw), (j, i)
        assert st.num_eval == stw.num_eval and list(st.level_regions) == list(stw.level_regions)


def test_queue_depth_is_two_and_needs_fixed_counts(mods):
    import torch
    ffi, synth, HipAZNet = mods
    net = HipAZNet(synth.make_head(seed=77, **synth.SMALL_DIMS), name="q_depth")
    m = torch.from_numpy(synth.make_feature_map(3, synth.SMALL_DIMS["C"], 38, 63)).cuda()
    p = ffi.AzContext.make_params(600, 1000, 1.0, 0.0)
    net.ctx.propose_launch(p, fmap=m)
    net.ctx.propose_launch(p, fmap=m)
    net.ctx.propose_launch(p, fmap=m)               # (three per lane since round 5: a search's result may trail it by a whole image)
    with pytest.raises(ffi.AzError):
        net.ctx.propose_launch(p, fmap=m)           # a fourth one: fetch first
    a = net.ctx.propose_fetch(want_scores=True)
    b = net.ctx.propose_fetch(want_scores=True)
    c3 = net.ctx.propose_fetch(want_scores=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[0], c3[0]) and np.array_equal(a[1], c3[1])
    with pytest.raises(ffi.AzError):
        net.ctx.propose_fetch()
    # a search with a data-dependent proposal count cannot have another queued behind it, nor be queued behind one
    pv = ffi.AzContext.make_params(600, 1000, 1.0, 0.0, fixed_num=False, Tc=0.4)
    net.ctx.propose_launch(pv, fmap=m)
    with pytest.raises(ffi.AzError):
        net.ctx.propose_launch(p, fmap=m)
    v = net.ctx.propose_fetch(want_scores=True)
    assert v[0].shape[0] > 0 and np.all(v[1] >= np.float32(0.4))
    net.ctx.propose_launch(p, fmap=m)
    with pytest.raises(ffi.AzError):
        net.ctx.propose_launch(pv, fmap=m)
    c = net.ctx.propose_fetch(want_scores=True)
    assert np.array_equal(a[0], c[0])
    # the candidate list of a fetched search is only readable while nothing has been queued behind it
    net.ctx.propose_launch(p, fmap=m)
    net.ctx.propose_launch(p, fmap=m)
    net.ctx.propose_fetch()
    with pytest.raises(ffi.AzError):
        net.ctx.last_candidates()
    net.ctx.propose_fetch()
    assert net.ctx.last_candidates()[0].shape[0] > 300


@pytest.mark.parametrize("kind", ["nan_premise", "whole_tree_miss"])
def test_rerun_under_graph_replay_with_a_staged_record(mods, kind):
    """A search that has to be repeated in another form inside propose_fetch (err bit 32: a NaN zoom score breaks the
    one-pass premise; err bit 256: a pruned tree needs a window the whole-tree pass lacks), with hipGraph replay ON and
    its result record staged into a caller's device buffer: the record is restaged by the rerun and equals the fetched
    result, which equals the plain search."""
    import torch
    from aznet_hip import dist as azdist
    ffi, synth, HipAZNet = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    if kind == "nan_premise":
        head["bz"] = np.full(1, np.nan, dtype=np.float32)
    fmap = torch.from_numpy(synth.make_feature_map(5, synth.SMALL_DIMS["C"], 38, 63)).cuda()
    k = 300
    ref = HipAZNet(head, name="g_ref")
    ref.set_conv(fmap)
    if kind == "nan_premise":
        p = ffi.AzContext.make_params(600, 1000, 1.0, 0.0, num_proposals=k)                  # one-pass plan -> level loop
        pref = ffi.AzContext.make_params(600, 1000, 1.0, 0.0, num_proposals=k, static_tree=False, full_spec=False)
    else:
        ref.propose(ffi.AzContext.make_params(600, 1000, 1.0, 0.0, tune=True))
        z = np.sort(ref.ctx.last_anchors()[1].astype(np.float64))
        Tz = float(z[int(0.3 * (len(z) - 1))])
        p = ffi.AzContext.make_params(600, 1000, 1.0, Tz, num_proposals=k, static_tree=False, full_spec=True)
        pref = ffi.AzContext.make_params(600, 1000, 1.0, Tz, num_proposals=k, static_tree=False, full_spec=False,
                                         pair_spec=False)
    Yw, Sw = ref.propose(pref, want_scores=True)
    net = HipAZNet(head, name="g_run")
    net.ctx.set_graphs(True)
    layout = ffi.AzContext.result_record_layout(k)
    buf = torch.zeros(layout[0], dtype=torch.uint8, device="cuda")
    for rnd in range(3):                       # (first un-captured, then captured, then replayed)
        buf.zero_()
        net.ctx.propose_launch(p, fmap=fmap, producer_done=True)
        net.ctx.stage_result(buf.data_ptr(), layout[0])
        Y, S = net.ctx.propose_fetch(want_scores=True)
        assert np.array_equal(Y, Yw) and np.array_equal(S, Sw), (kind, rnd)
        torch.cuda.synchronize()
        rec = azdist.unpack_device_record(buf.cpu().numpy(), layout, k)
        assert rec is not None and np.array_equal(rec[0], Yw) and np.array_equal(rec[1], Sw), (kind, rnd)


# ---- two lanes (az_set_lanes): queued searches take turns between two streams of ONE context and overlap on the GPU --------
@pytest.mark.parametrize("nan_zoom", [False, True])
def test_two_lanes_equal_one_lane_bit_for_bit(mods, nan_zoom):
    """Same sequence as test_queue_ahead_equals_alternating with az_set_lanes(2): mixed shapes, forms and reruns (NaN zoom
    under the one-pass plan; a level that outgrows the fused kernels), results oldest first, equal to the one-at-a-time
    context.  Then depth: each lane queues up to three searches, so six may be pending."""
    import torch
    ffi, synth, HipAZNet = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    if nan_zoom:
        head["bz"] = np.full(1, np.nan, dtype=np.float32)
    cases = _cases(ffi, synth)
    maps = [torch.from_numpy(m).cuda() for _, m in cases]
    ref_net = HipAZNet(head, name="l_ref")
    want = []
    for (p, _), m in zip(cases, maps):
        ref_net.set_conv(m)
        want.append(ref_net.propose(p, want_scores=True, want_stats=True))
    net = HipAZNet(head, name="l_two")
    net.ctx.set_lanes(2)
    seq = list(range(len(cases))) * 3
    for depth in (2, 4, 6):
        got, q = [], 0
        for j in range(len(seq) + depth - 1):
            if j < len(seq):
                net.ctx.propose_launch(cases[seq[j]][0], fmap=maps[seq[j]])
                q += 1
            if q == depth or j >= len(seq):
                got.append(net.ctx.propose_fetch(want_scores=True, want_stats=True))
                q -= 1
        assert len(got) == len(seq)
        for j, i in enumerate(seq):
            Y, S, st = got[j]
            Yw, Sw, stw = want[i]
            assert np.array_equal(Y, Yw) and np.array_equal(S, Sw), (depth, j, i)
            assert st.num_eval == stw.num_eval and list(st.level_regions) == list(stw.level_regions)
    with pytest.raises(ffi.AzError):
        net.ctx.propose_fetch()


def test_two_lanes_candidates_sync_calls_and_lane_switching(mods):
    import torch
    ffi, synth, HipAZNet = mods
    net = HipAZNet(synth.make_head(seed=77, **synth.SMALL_DIMS), name="l_misc")
    maps = [torch.from_numpy(synth.make_feature_map(3 + i, synth.SMALL_DIMS["C"], 38, 63)).cuda() for i in range(3)]
    p = ffi.AzContext.make_params(600, 1000, 1.0, 0.2, static_tree=False)
    ref = []
    for m in maps:
        net.set_conv(m)
        Y, S = net.propose(p, want_scores=True)
        ref.append((Y, S) + net.ctx.last_candidates())
    net.ctx.set_lanes(2)
    # the candidate list of the search fetched last, whichever lane ran it
    for rnd in range(2):
        for i, m in enumerate(maps):
            net.ctx.propose_launch(p, fmap=m)
            Y, S = net.ctx.propose_fetch(want_scores=True)
            Ya, Sa = net.ctx.last_candidates()
            assert np.array_equal(Y, ref[i][0]) and np.array_equal(Ya, ref[i][2]) and np.array_equal(Sa, ref[i][3]), (rnd, i)
    # a synchronous call between queued ones is refused while something is queued, fine otherwise (first lane)
    net.ctx.propose_launch(p, fmap=maps[0])
    with pytest.raises(ffi.AzError):
        net.propose(p)
    with pytest.raises(ffi.AzError):
        net.ctx.set_lanes(1)
    net.ctx.propose_fetch()
    net.set_conv(maps[1])
    Y, S = net.propose(p, want_scores=True)
    assert np.array_equal(Y, ref[1][0]) and np.array_equal(net.ctx.last_candidates()[0], ref[1][2])
    # the map set on the context itself (not handed over with the launch) is read by whichever lane takes the search
    net.set_conv(maps[2])
    net.ctx.propose_launch(p)
    net.ctx.propose_launch(p)
    a = net.ctx.propose_fetch(want_scores=True)
    b = net.ctx.propose_fetch(want_scores=True)
    assert np.array_equal(a[0], ref[2][0]) and np.array_equal(b[0], ref[2][0]) and np.array_equal(b[1], ref[2][1])
    # kernel times of both lanes come back together
    net.ctx.set_profiling(2 | 4)
    net.ctx.propose_launch(p, fmap=maps[0]); net.ctx.propose_launch(p, fmap=maps[1])
    net.ctx.propose_fetch(); net.ctx.propose_fetch()
    kt = net.ctx.last_kernel_times()
    net.ctx.set_profiling(0)
    assert sum(1 for n, l, ms in kt if n == "final_select" or n == "select") == 2
    # back to one lane, a new head: the second lane goes away and comes back
    net.ctx.set_lanes(1)
    net.ctx.propose_launch(p, fmap=maps[0]); net.ctx.propose_launch(p, fmap=maps[1])
    assert np.array_equal(net.ctx.propose_fetch(), ref[0][0]) and np.array_equal(net.ctx.propose_fetch(), ref[1][0])
    net.ctx.set_lanes(2)
    net.ctx.load_head(synth.make_head(seed=77, **synth.SMALL_DIMS))
    net.ctx.propose_launch(p, fmap=maps[0]); net.ctx.propose_launch(p, fmap=maps[1]); net.ctx.propose_launch(p, fmap=maps[2])
    for i in range(3):
        assert np.array_equal(net.ctx.propose_fetch(), ref[i][0])


@pytest.mark.parametrize("kind", ["nan_premise", "whole_tree_miss"])
def test_two_lanes_rerun_with_a_staged_record(mods, kind):
    """test_rerun_under_graph_replay_with_a_staged_record on both lanes: the search that is repeated inside propose_fetch and
    its restaged record, whichever lane it ran on, with another search queued on the other lane."""
    import torch
    from aznet_hip import dist as azdist
    ffi, synth, HipAZNet = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    if kind == "nan_premise":
        head["bz"] = np.full(1, np.nan, dtype=np.float32)
    fmap = torch.from_numpy(synth.make_feature_map(5, synth.SMALL_DIMS["C"], 38, 63)).cuda()
    k = 300
    ref = HipAZNet(head, name="lg_ref")
    ref.set_conv(fmap)
    if kind == "nan_premise":
        p = ffi.AzContext.make_params(600, 1000, 1.0, 0.0, num_proposals=k)
        pref = ffi.AzContext.make_params(600, 1000, 1.0, 0.0, num_proposals=k, static_tree=False, full_spec=False)
    else:
        ref.propose(ffi.AzContext.make_params(600, 1000, 1.0, 0.0, tune=True))
        z = np.sort(ref.ctx.last_anchors()[1].astype(np.float64))
        Tz = float(z[int(0.3 * (len(z) - 1))])
        p = ffi.AzContext.make_params(600, 1000, 1.0, Tz, num_proposals=k, static_tree=False, full_spec=True)
        pref = ffi.AzContext.make_params(600, 1000, 1.0, Tz, num_proposals=k, static_tree=False, full_spec=False,
                                         pair_spec=False)
    Yw, Sw = ref.propose(pref, want_scores=True)
    net = HipAZNet(head, name="lg_run")
    net.ctx.set_lanes(2)
    net.ctx.set_graphs(True)
    layout = ffi.AzContext.result_record_layout(k)
    bufs = [torch.zeros(layout[0], dtype=torch.uint8, device="cuda") for _ in range(2)]
    for rnd in range(4):
        for b in bufs:
            b.zero_()
        for b in bufs:
            net.ctx.propose_launch(p, fmap=fmap, producer_done=True)
            net.ctx.stage_result(b.data_ptr(), layout[0])
        for b in bufs:
            Y, S = net.ctx.propose_fetch(want_scores=True)
            assert np.array_equal(Y, Yw) and np.array_equal(S, Sw), (kind, rnd)
        torch.cuda.synchronize()
        for b in bufs:
            rec = azdist.unpack_device_record(b.cpu().numpy(), layout, k)
            assert rec is not None and np.array_equal(rec[0], Yw) and np.array_equal(rec[1], Sw), (kind, rnd)


@pytest.mark.parametrize("depth", [3, 4])
@pytest.mark.parametrize("kind", ["host", "dev_nchw"])
def test_two_lanes_with_maps_set_on_the_context(mods, depth, kind):
    """az_set_feature_map_* on the CONTEXT, then az_propose_launch (no map in the call), two lanes, searches queued ahead,
    every search another map, long dense trees on one lane beside short ones on the other: the context keeps two
    channel-last copies and writes them in turn, so the second lane takes a private copy of a map it is handed that way --
    a lane-1 search that is fetched (or even starts) after the context has seen two more maps must still read ITS map."""
    import torch
    ffi, synth, HipAZNet = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    H, W, scale = 600, 1000, 1.0
    C = synth.SMALL_DIMS["C"]
    maps = [synth.make_scene_map(200 + i, C, 38, 63) for i in range(10)]
    ref = HipAZNet(head, name="lanes_ctxmap_ref")
    ref.set_conv(maps[0])
    z = np.sort(ref.ctx.head_forward(np.hstack([np.zeros((8, 1)), ref.ctx.divide_region(
        np.array([[0.0, 0.0, W - 1.0, H - 1.0]]), 10.0) * scale]).astype(np.float32))[0].ravel())
    hi = float(0.5 * (z[-1] + 1.0))                        # nearly every tree ends with the root's children
    # dense trees (Tz = 0, level by level) on the odd positions = the second lane, sparse ones on the first
    seq = [(i % len(maps), 0.0 if i % 2 == 1 else hi) for i in range(24)]
    want = []
    for mi, tz in seq:
        ref.set_conv(maps[mi])
        want.append(ref.propose(ffi.AzContext.make_params(H, W, scale, tz, static_tree=False), want_scores=True))
    net = HipAZNet(head, name="lanes_ctxmap")
    net.ctx.set_lanes(2)
    tmaps = [torch.from_numpy(m).cuda() for m in maps] if kind == "dev_nchw" else None

    def launch(j):
        mi, tz = seq[j]
        if kind == "host":
            net.set_conv(maps[mi])                        # az_set_feature_map_host
        else:
            net.set_conv(tmaps[mi], wait=False)           # az_set_feature_map_dev_async: the transpose is only enqueued
        net.ctx.propose_launch(ffi.AzContext.make_params(H, W, scale, tz, static_tree=False))
    got, launched = [], 0
    for i in range(len(seq)):
        while launched < min(len(seq), i + depth):
            launch(launched)
            launched += 1
        got.append(net.ctx.propose_fetch(want_scores=True))
    for i, ((Y, S), (Yw, Sw)) in enumerate(zip(got, want)):
        assert np.array_equal(Y, Yw) and np.array_equal(S, Sw), (i, seq[i])


@pytest.mark.parametrize("depth", [1, 2, 3])
def test_staged_searches_on_one_lane(mods, depth):
    """One-lane contexts enqueue a search of one head pass in stages on streams of their own (round 5: RoIPool + int6 + slab sum
    | int7 + heads | geometry + selection + result copy).  A sequence that mixes staged searches (the one-pass plan, the
    whole-tree pass after two full trees, the closure) with searches in other forms (sparse trees level by level, the plain
    loop, a data-dependent proposal count), two image shapes, different maps, unit calls in between and records staged for the
    exchange, queued up to three deep: every result as the one-at-a-time reference gives it."""
    import torch
    ffi, synth, HipAZNet = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    C = synth.SMALL_DIMS["C"]
    shapes = [(600, 1000, 1.0), (375, 500, 1.6)]
    maps = {}
    for si, (H, W, sc) in enumerate(shapes):
        fh, fw = synth.conv_out_size(int(round(H * sc))), synth.conv_out_size(int(round(W * sc)))
        maps[si] = [synth.make_scene_map(400 + 10 * si + j, C, fh, fw) for j in range(3)]
    ref = HipAZNet(head, name="staged_ref")
    net = HipAZNet(head, name="staged")
    ref.set_conv(maps[0][0])
    z = np.sort(ref.ctx.head_forward(np.hstack([np.zeros((8, 1)), ref.ctx.divide_region(
        np.array([[0.0, 0.0, 999.0, 599.0]]), 10.0)]).astype(np.float32))[0].ravel())
    sparse = float(0.5 * (z[3] + z[4]))
    mk = ffi.AzContext.make_params
    kinds = [dict(Tz=0.0),                                            # one-pass plan: two stages
             dict(Tz=0.0, static_tree=False),                         # level loop, then (history: full trees) the whole-tree pass
             dict(Tz=0.0, static_tree=False),
             dict(Tz=0.0, static_tree=False),
             dict(Tz=sparse, static_tree=False),                      # a pruned tree: other forms, all on the context's stream
             dict(Tz=0.0, static_tree=False, full_spec="closure"),    # closure pass: three stages
             dict(Tz=0.0, static_tree=False, full_spec=True),
             dict(Tz=sparse, static_tree=False, speculate=False, fused=False, fused_levels=False, pair_spec=False, full_spec=False),
             dict(Tz=0.0)]
    seq = []
    for rnd in range(3):
        for ki, kw in enumerate(kinds):
            si = (ki + rnd) % 2
            seq.append((si, (ki + rnd) % 3, kw))
    want = []
    for si, mi, kw in seq:
        H, W, sc = shapes[si]
        ref.set_conv(maps[si][mi])
        want.append(ref.propose(mk(H, W, sc, kw["Tz"], static_tree=False, speculate=False, fused=False, fused_levels=False,
                                   pair_spec=False, full_spec=False, early_end=False), want_scores=True))
    tmaps = {si: [torch.from_numpy(m).cuda().contiguous(memory_format=torch.channels_last) for m in maps[si]] for si in maps}
    layout = ffi.AzContext.result_record_layout(300)
    stage = torch.zeros((len(seq), layout[0]), dtype=torch.uint8, device="cuda")
    got, launched = [], 0
    for i in range(len(seq)):
        while launched < min(len(seq), i + depth):
            si, mi, kw = seq[launched]
            H, W, sc = shapes[si]
            kw2 = dict(kw)
            tz = kw2.pop("Tz")
            net.ctx.propose_launch(mk(H, W, sc, tz, **kw2), fmap=tmaps[si][mi], producer_done=True)
            net.ctx.stage_result(stage[launched].data_ptr(), layout[0])
            launched += 1
        got.append(net.ctx.propose_fetch(want_scores=True, want_stats=True))
        if i % 5 == 4 and launched == i + 1:
            # a unit call between searches (nothing queued): works in the per-search buffers
            assert np.array_equal(net.ctx.divide_region(np.array([[0.0, 0.0, 999.0, 599.0]]), 10.0),
                                  ref.ctx.divide_region(np.array([[0.0, 0.0, 999.0, 599.0]]), 10.0))
    torch.cuda.synchronize()
    raw = stage.cpu().numpy()
    from aznet_hip import dist as azdist
    forms = set()
    for i, ((Y, S, st), (Yw, Sw)) in enumerate(zip(got, want)):
        assert np.array_equal(Y, Yw) and np.array_equal(S, Sw), (i, seq[i][2], int(st.search_form))
        rec = azdist.unpack_device_record(raw[i], layout, 300)
        assert np.array_equal(rec[0], Yw) and np.array_equal(rec[1], Sw), ("staged record", i)
        forms.add(int(st.search_form))
    assert {2, 3, 4} <= forms, forms                       # the whole-tree pass, the closure pass and the one-pass plan all ran
