"""Early end of a sparse search: a search whose context's previous search of the image shape had no regions from some level
on is enqueued only up to that level; regions there after all -> the search is run again in full.  Same bits as the plain
level loop either way (lib/detect/test.py:373-391: the reference's loop simply finds Z empty)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from aznet_hip import ffi, synth
    from aznet_hip.net import HipAZNet
    return ffi, synth, HipAZNet


def _plain(ffi, H, W, scale, Tz, **kw):
    return ffi.AzContext.make_params(H, W, scale, Tz, speculate=False, fused=False, fused_levels=False, static_tree=False,
                                     pair_spec=False, full_spec=False, early_end=False, **kw)


@pytest.mark.parametrize("H,W,scale", [(600, 1000, 1.0), (375, 500, 1.6), (480, 640, 1.25)])
@pytest.mark.parametrize("lanes", [1, 2])
def test_sparse_searches_end_early_and_a_deeper_tree_is_run_again(mods, H, W, scale, lanes):
    ffi, synth, HipAZNet = mods
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    net = HipAZNet(head, name="early_end")
    ref = HipAZNet(head, name="early_end_ref")
    net.ctx.set_lanes(lanes)
    C = synth.SMALL_DIMS["C"]
    fh, fw = int(np.ceil(H * scale / 16.0)), int(np.ceil(W * scale / 16.0))
    maps = [synth.make_feature_map(40 + i, C, fh, fw) for i in range(3)]
    # thresholds from the zoom scores of the first map: one that prunes everything below level 2, one that keeps most
    ref.set_conv(maps[0])
    _, _, st0 = ref.propose(_plain(ffi, H, W, scale, 0.0), want_scores=True, want_stats=True)
    assert st0.n_levels >= 4
    # hi: above every level-2 zoom score of the three maps and below the root's forced 1.0 (test.py:383-384): trees [1, 8]
    B1 = ref.ctx.divide_region(np.array([[0.0, 0.0, W - 1.0, H - 1.0]]), 10.0)
    zmax = 0.0
    for m in maps:
        ref.set_conv(m)
        z, _, _ = ref.ctx.head_forward(np.hstack([np.zeros((len(B1), 1)), B1 * scale]).astype(np.float32))
        zmax = max(zmax, float(z.max()))
    assert zmax < 1.0
    hi, lo = 0.5 * (zmax + 1.0), 0.0

    def run(Tz, fmap):
        net.set_conv(fmap)
        ref.set_conv(fmap)
        Y, S, st = net.propose(ffi.AzContext.make_params(H, W, scale, Tz, static_tree=False, full_spec=False),
                               want_scores=True, want_stats=True)
        Yr, Sr, sr = ref.propose(_plain(ffi, H, W, scale, Tz), want_scores=True, want_stats=True)
        assert np.array_equal(Y, Yr) and np.array_equal(S, Sr)
        assert list(st.level_regions) == list(sr.level_regions) and st.n_candidates == sr.n_candidates
        assert st.num_eval == sr.num_eval and st.depth == sr.depth
        Ya, Sa = net.ctx.last_candidates()
        Yb, Sb = ref.ctx.last_candidates()
        assert np.array_equal(Ya, Yb) and np.array_equal(Sa, Sb)
        return st

    s1 = run(hi, maps[0])                                # no history yet: every level enqueued
    assert list(s1.level_regions[:2]) == [1, len(B1)] and sum(s1.level_regions[2:s1.n_levels]) == 0 and s1.n_reruns == 0
    assert s1.n_passes == 1 and s1.pass_rows[0] > 1 + len(B1)
    for k in range(6):                                   # seven searches that ended after their second level ...
        sk = run(hi, maps[(k + 1) % 3])
        assert sk.n_reruns == 0 and sk.pass_rows[0] > 1 + len(B1)      # (fewer than 7 of the last 8: not cut yet)
    s2 = run(hi, maps[1])                                # ... 7 of the last 8: this one is enqueued only that far
    assert s2.n_reruns == 0 and s2.n_passes == 1 and list(s2.pass_rows[:1]) == [1 + len(B1)]
    assert int(s2.pass_levels[0]) == 3                   # (the pass evaluated levels 1 and 2)
    # the same with a data-dependent proposal count (cfg.SEAR.FIXED_PROPOSAL_NUM = False: everything with score >= Tc)
    net.set_conv(maps[0]); ref.set_conv(maps[0])
    Yt, St = net.propose(ffi.AzContext.make_params(H, W, scale, hi, static_tree=False, full_spec=False, fixed_num=False, Tc=0.3),
                         want_scores=True)
    Yq, Sq = ref.propose(_plain(ffi, H, W, scale, hi, fixed_num=False, Tc=0.3), want_scores=True)
    assert np.array_equal(Yt, Yq) and np.array_equal(St, Sq)
    s3 = run(lo, maps[2])                                # a full tree behind sparse ones: cut, missed, run again
    assert s3.n_reruns == 1 and sum(s3.level_regions[2:s3.n_levels]) > 0
    s4 = run(hi, maps[0])                                # (still 7 sparse searches among the last 8: cut, and right)
    assert s4.n_reruns == 0 and list(s4.pass_rows[:s4.n_passes]) == [1 + len(B1)]
    s5 = run(lo, maps[1])                                # ... cut, and wrong once more
    assert s5.n_reruns == 1
    for k in range(6):                                   # from here on 6 or fewer of the last 8 are sparse: alternating sparse /
        sk = run(hi if k % 2 == 0 else lo, maps[k % 3])  # dense images are never cut, nothing is run twice
        assert sk.n_reruns == 0 and sk.pass_rows[0] > 1 + len(B1)


@pytest.mark.parametrize("lanes,depth", [(1, 2), (2, 3), (2, 4)])
def test_an_early_end_that_misses_is_run_again_inside_a_full_queue(mods, lanes, depth):
    """The same with searches queued ahead (az_propose_launch x depth before the first fetch): the rerun of a search whose tree
    went deeper than the cut happens inside az_propose_fetch with other searches queued behind it, on one lane and on two."""
    ffi, synth, HipAZNet = mods
    H, W, scale = 600, 1000, 1.0
    head = synth.make_head(seed=77, **synth.SMALL_DIMS)
    net = HipAZNet(head, name="early_end_q")
    ref = HipAZNet(head, name="early_end_q_ref")
    net.ctx.set_lanes(lanes)
    C = synth.SMALL_DIMS["C"]
    maps = [synth.make_feature_map(60 + i, C, 38, 63) for i in range(4)]
    import torch
    tmaps = [torch.from_numpy(m).cuda().contiguous(memory_format=torch.channels_last) for m in maps]
    B1 = ref.ctx.divide_region(np.array([[0.0, 0.0, W - 1.0, H - 1.0]]), 10.0)
    zmax = 0.0
    for m in maps:
        ref.set_conv(m)
        z, _, _ = ref.ctx.head_forward(np.hstack([np.zeros((len(B1), 1)), B1 * scale]).astype(np.float32))
        zmax = max(zmax, float(z.max()))
    hi = 0.5 * (zmax + 1.0)
    # a run of sparse trees (each lane's last four end after their second level: cut), a dense one (missed, run again), a mix
    # (a lane cuts when 7 of ITS last 8 searches ended early: with two lanes every other search of the sequence)
    seq = [hi] * 18 + [0.0, hi, 0.0, 0.0] + [hi] * 18 + [0.0, hi]
    prm = [ffi.AzContext.make_params(H, W, scale, t, static_tree=False, full_spec=False) for t in seq]
    want = []
    for i, t in enumerate(seq):
        ref.set_conv(maps[i % 4])
        want.append(ref.propose(_plain(ffi, H, W, scale, t), want_scores=True))
    got, reruns, launched = [], 0, 0
    for i in range(len(seq)):
        while launched < min(len(seq), i + depth):
            net.ctx.propose_launch(prm[launched], fmap=tmaps[launched % 4], producer_done=True)
            launched += 1
        Y, S, st = net.ctx.propose_fetch(want_scores=True, want_stats=True)
        got.append((Y, S))
        reruns += int(st.n_reruns)
    for i, ((Y, S), (Yr, Sr)) in enumerate(zip(got, want)):
        assert np.array_equal(Y, Yr) and np.array_equal(S, Sr), i
    assert reruns >= 1                                    # (the first dense search behind the sparse ones)
