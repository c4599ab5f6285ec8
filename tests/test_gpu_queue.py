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
    with pytest.raises(ffi.AzError):
        net.ctx.propose_launch(p, fmap=m)           # a third one: fetch first
    a = net.ctx.propose_fetch(want_scores=True)
    b = net.ctx.propose_fetch(want_scores=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
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
