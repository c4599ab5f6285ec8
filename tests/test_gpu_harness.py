"""The harness / CLI row: `test_proposals`, `test_net_shared`,
`tools/prop_az.py`, the `.caffemodel` path and the VGG16 plumbing, on the GPU.

Pinned by tests/golden/g13_harness.npz -- what the REFERENCE's own test_proposals / test_net_shared
(lib/detect/test.py:486-539, 670-778) printed and pickled for a 2-image stub imdb with the seed-77 / seed-99
small heads on the CPU (oracle/gen_golden_harness.py)."""
import io
import os
import pickle
import re
import shutil
import subprocess
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest

from helpers import load

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(REPO, "az-net_amd", "tools")


def scrub(text):
    return re.sub(r"\d+\.\d{3}s", "0.000s", text)


@pytest.fixture(scope="module")
def mods():
    import torch
    from aznet_hip import ffi, synth
    from aznet_hip.net import HipAZNet, HipDetNet
    from oracle import az_oracle as orc
    return torch, ffi, synth, HipAZNet, HipDetNet, orc


@pytest.fixture()
def harness_cfg():
    """cfg as gen_golden_harness.py set it; restored afterwards."""
    from detect import config as C
    old = (C.cfg.TEST.NUM_PROPOSALS, C.cfg.SEAR.get("Tz", 0.0), C.cfg.EXP_DIR, C.cfg.TEST.MAX_SIZE,
           C.cfg.SEAR.BATCH_SIZE)
    C.cfg.TEST.MAX_SIZE = 1000
    C.cfg.SEAR.BATCH_SIZE = 10000
    C.cfg.TEST.NUM_PROPOSALS = 100
    C.cfg_set_path("harness_test")
    C.cfg_set_mode("Test", 0.0)
    yield C
    C.cfg.TEST.NUM_PROPOSALS = old[0]
    C.cfg_set_mode("Test", old[1])
    C.cfg.EXP_DIR = old[2]
    C.cfg.TEST.MAX_SIZE, C.cfg.SEAR.BATCH_SIZE = old[3], old[4]
    shutil.rmtree(os.path.join(C.cfg.ROOT_DIR, "output", "harness_test"), ignore_errors=True)


class _MapBackbone(object):
    """Stands where VGG16 would: hands back the seeded conv5_3 of the image the imdb served last (the golden
    run's 'full' net did the same), as a CUDA tensor."""

    def __init__(self, torch, synth, C, fh, fw):
        self.torch, self.synth, self.C, self.fh, self.fw = torch, synth, C, fh, fw
        self.device = torch.device("cuda", 0)
        self.cur = 0
        self.served = []          # images the imdb handed out, oldest first (the harness may read ahead of the backbone)

    def __call__(self, blob):
        assert tuple(blob.shape[:2]) == (1, 3)
        if self.served:
            self.cur = self.served.pop(0)
        m = self.synth.make_feature_map(40 + self.cur, self.C, self.fh, self.fw)
        return self.torch.from_numpy(m).to(self.device)


def _stub_imdb(synth, g, backbone):
    from datasets.imdb import imdb as imdb_base
    H, W, n = int(g["H"]), int(g["W"]), int(g["n_img"])

    class Stub(imdb_base):
        def __init__(self):
            imdb_base.__init__(self, "stub_2img")
            self._image_index = list(range(n))
            self._classes = ["c%d" % i for i in range(21)]
            self.nms_dets = None

        def image_at(self, i):
            backbone.served.append(i)
            return synth.make_image(i, H, W)

        def image_path_at(self, i):
            return "synthetic:/%d" % i

        def evaluate_detections(self, nms_dets, output_dir):
            self.nms_dets, self.eval_dir = nms_dets, output_dir
    return Stub()


@pytest.fixture()
def rig(mods, harness_cfg):
    torch, ffi, synth, HipAZNet, HipDetNet, orc = mods
    g = load("g13_harness.npz")
    H, W, scale = int(g["H"]), int(g["W"]), float(g["scale"])
    fh, fw = synth.conv_out_size(int(round(H * scale))), synth.conv_out_size(int(round(W * scale)))
    bb = _MapBackbone(torch, synth, synth.SMALL_DIMS["C"], fh, fw)
    net = HipAZNet(synth.make_head(seed=77, **synth.SMALL_DIMS), backbone=bb, name="az_small")
    dnet = HipDetNet(synth.make_det_head(seed=99, **synth.SMALL_DET_DIMS), net)
    return g, net, dnet, _stub_imdb(synth, g, bb), harness_cfg


def test_test_proposals_matches_the_reference_run(rig, mods):
    """proposals.pkl: same dict layout, path rule, printed lines and (within the fp32 head tolerance) boxes
    as the reference's test_proposals; and equal, bit for bit, to per-image im_propose."""
    torch, ffi, synth, HipAZNet, HipDetNet, orc = mods
    g, net, dnet, imdb, C = rig
    from detect import test as T
    buf = io.StringIO()
    with redirect_stdout(buf):
        prop_file = T.test_proposals({"full": net, "fc": net}, imdb)
    assert os.path.relpath(prop_file, C.cfg.ROOT_DIR) == str(g["prop_relpath"]).replace("/harness/", "/harness_test/")
    assert scrub(buf.getvalue()) == str(g["prop_stdout"])          # per-image line, depth/eval counts, summary
    with open(prop_file, "rb") as f:
        prop = pickle.load(f)
    assert sorted(prop.keys()) == sorted(str(k) for k in g["prop_keys"]) == ["boxes", "recall", "time"]
    assert isinstance(prop["time"], float) and prop["time"] > 0 and prop["recall"] == int(g["prop_recall"]) == 0
    assert isinstance(prop["boxes"], list) and len(prop["boxes"]) == int(g["n_img"])
    for i, b in enumerate(prop["boxes"]):
        ref = g["prop_boxes%d" % i]
        assert b.dtype == np.float64 and b.shape == ref.shape == (100, 4)
        # the reference's head ran on the CPU: same proposals within 1e-4-driven tolerances, except where
        # two candidates' scores tie within the tolerance at the cut
        hit = [np.abs(b - r).max(axis=1).min() <= 1e-3 for r in ref]
        assert np.mean(hit) >= 0.97, (i, np.mean(hit))
        with redirect_stdout(io.StringIO()):
            again = T.im_propose(net, imdb.image_at(i))
        assert np.array_equal(again, b)


@pytest.mark.parametrize("nb", [2, 3, 16])
def test_test_proposals_in_lockstep_batches(rig, mods, nb):
    """cfg.TEST.BATCH_IMAGES: consecutive images searched in lockstep -- the printed lines (the reference run's, g13) and every
    image's boxes are those of the one-by-one loop, bit for bit."""
    torch, ffi, synth, HipAZNet, HipDetNet, orc = mods
    g, net, dnet, imdb, C = rig
    from detect import test as T
    with redirect_stdout(io.StringIO()):
        with open(T.test_proposals({"full": net, "fc": net}, imdb), "rb") as f:
            one_by_one = pickle.load(f)
    C.cfg.TEST.BATCH_IMAGES = nb
    try:
        buf = io.StringIO()
        with redirect_stdout(buf):
            prop_file = T.test_proposals({"full": net, "fc": net}, imdb)
    finally:
        C.cfg.TEST.BATCH_IMAGES = 1
    assert scrub(buf.getvalue()) == str(g["prop_stdout"])
    with open(prop_file, "rb") as f:
        prop = pickle.load(f)
    assert len(prop["boxes"]) == int(g["n_img"]) == len(one_by_one["boxes"])
    for a, b in zip(prop["boxes"], one_by_one["boxes"]):
        assert a.dtype == np.float64 and np.array_equal(a, b)


def test_test_net_shared_matches_the_reference_run(rig, mods):
    """detections.pkl / evaluate_detections input of test_net_shared (BASELINE config 3)."""
    torch, ffi, synth, HipAZNet, HipDetNet, orc = mods
    g, net, dnet, imdb, C = rig
    from detect import test as T
    n = int(g["n_img"])
    # per-image results of the path itself, recorded while the harness runs
    # (at the Fast R-CNN head's forward: the dataset loop enqueues image i+1 before image i's bookkeeping and does not go
    #  through im_detect_shared)
    rec = []
    inner = T._frcnn_forward

    def recording(*a, **kw):
        s, b, c = inner(*a, **kw)
        rec.append((s.copy(), b.copy()))
        return s, b, c
    T._frcnn_forward = recording
    try:
        buf = io.StringIO()
        with redirect_stdout(buf):
            nms_dets = T.test_net_shared({"full": net, "fc": net}, {"fc": dnet}, imdb)
    finally:
        T._frcnn_forward = inner
    assert scrub(buf.getvalue()) == str(g["det_stdout"])
    det_file = os.path.join(C.get_output_dir(imdb, net), "detections.pkl")
    assert os.path.relpath(det_file, C.cfg.ROOT_DIR) == str(g["det_relpath"]).replace("/harness/", "/harness_test/")
    assert imdb.eval_dir == os.path.dirname(det_file) and imdb.nms_dets is nms_dets
    with open(det_file, "rb") as f:
        all_boxes = pickle.load(f)
    assert len(all_boxes) == 21 and len(all_boxes[0]) == n and len(rec) == n
    # (1) the path's per-image outputs vs the reference run's (CPU heads): 1e-4
    for i in range(n):
        s, b = rec[i]
        assert s.dtype == np.float64 and s.shape == g["det_scores%d" % i].shape
        # rows are ordered by proposal rank, which may swap where two proposal scores tie within the head
        # tolerance (and the 100th proposal may differ): row-wise agreement for all but a few rows
        close_s = np.abs(s - g["det_scores%d" % i]).max(axis=1) <= 1e-4
        close_b = np.abs(b - g["det_boxes%d" % i]).max(axis=1) <= 2e-2
        assert close_s.mean() >= 0.9 and close_b.mean() >= 0.9, (close_s.mean(), close_b.mean())
    # (2) the harness bookkeeping == the oracle's restatement (pinned to the reference by the golden) applied
    #     to those per-image outputs: bit-exact lists, thresholds, NMS keep sets
    want, thresh = orc.net_shared_select(rec, 21)
    want_nms = orc.apply_nms(want, C.cfg.TEST.NMS)
    for j in range(1, 21):
        for i in range(n):
            assert all_boxes[j][i].dtype == np.float32 and np.array_equal(all_boxes[j][i], want[j][i])
            a, w = nms_dets[j][i], want_nms[j][i]
            assert (isinstance(a, list) and isinstance(w, list) and a == w == []) or np.array_equal(a, w)
            # and the same detections as the reference run, up to the head tolerance
            ref = g["det_all_%d_%d" % (j, i)]
            assert abs(all_boxes[j][i].shape[0] - ref.shape[0]) <= 5


@pytest.mark.parametrize("nb", [2, 16])
def test_test_net_shared_with_lockstep_proposals(rig, mods, nb):
    """cfg.TEST.BATCH_IMAGES in the detection loop: the proposals of consecutive images in lockstep batches, the Fast R-CNN
    head image by image -- the reference run's printed lines (g13), detections.pkl and NMS lists of the one-by-one loop."""
    torch, ffi, synth, HipAZNet, HipDetNet, orc = mods
    g, net, dnet, imdb, C = rig
    from detect import test as T

    def run():
        buf = io.StringIO()
        with redirect_stdout(buf):
            nms = T.test_net_shared({"full": net, "fc": net}, {"fc": dnet}, imdb)
        with open(os.path.join(C.get_output_dir(imdb, net), "detections.pkl"), "rb") as f:
            return buf.getvalue(), nms, pickle.load(f)
    out1, nms1, det1 = run()
    C.cfg.TEST.BATCH_IMAGES = nb
    try:
        out2, nms2, det2 = run()
    finally:
        C.cfg.TEST.BATCH_IMAGES = 1
    assert scrub(out2) == scrub(out1) == str(g["det_stdout"])
    n = int(g["n_img"])
    for j in range(1, 21):
        for i in range(n):
            assert np.array_equal(det1[j][i], det2[j][i])
            a, b = nms1[j][i], nms2[j][i]
            assert (isinstance(a, list) and isinstance(b, list) and a == b == []) or np.array_equal(a, b)


def test_lockstep_batches_over_a_dataset_of_mixed_shapes(mods):
    """A dataset mixes image shapes: test_proposals with cfg.TEST.BATCH_IMAGES puts images of DIFFERENT shapes into one lockstep
    batch (az_batch_launch_shapes: every image its own pre-pass, map size and clipping box), regroups only where the number
    of levels differs (the two small images), and still prints and stores everything in dataset order -- the lines and boxes
    of the one-by-one loop.  A narrow VGG16 reads the blobs (every map is its own image's)."""
    torch, ffi, synth, HipAZNet, HipDetNet, orc = mods
    from aznet_hip.backbone import VGG16Conv5
    from datasets.imdb import imdb as imdb_base
    from detect import config as C
    from detect import test as T
    shapes = [(375, 500), (600, 1000), (375, 500), (160, 240), (500, 375), (600, 1000), (375, 500), (333, 500), (500, 375),
              (375, 500), (150, 200), (375, 500), (375, 500), (500, 375)]
    ims = [synth.make_scene_image(700 + j, h, w) for j, (h, w) in enumerate(shapes)]

    class Mixed(imdb_base):
        def __init__(self):
            imdb_base.__init__(self, "mixed_shapes")
            self._image_index = list(range(len(ims)))
            self._classes = ["c%d" % i for i in range(21)]

        def image_at(self, i):
            return ims[i]

        def image_path_at(self, i):
            return "synthetic:/%d" % i
    old = (C.cfg.SEAR.get("Tz", 0.0), C.cfg.EXP_DIR, C.cfg.TEST.NUM_PROPOSALS)
    C.cfg_set_path("harness_mixed")
    C.cfg_set_mode("Test", 0.3)
    C.cfg.TEST.NUM_PROPOSALS = 100
    try:
        bb = VGG16Conv5(device="cuda:0", seed=11, width_div=32)
        net = HipAZNet(synth.make_head(seed=77, **synth.SMALL_DIMS), backbone=bb, name="mixed")
        bb.normalize_output(T._get_image_blob(ims[1], net)[0])
        imdb = Mixed()

        def run():
            buf = io.StringIO()
            with redirect_stdout(buf):
                pf = T.test_proposals({"full": net, "fc": net}, imdb)
            with open(pf, "rb") as f:
                return scrub(buf.getvalue()), pickle.load(f)["boxes"]
        out1, boxes1 = run()
        assert len({b.shape[0] for b in boxes1}) >= 1 and len(boxes1) == len(ims)
        for nb in (2, 4, 16):
            C.cfg.TEST.BATCH_IMAGES = nb
            try:
                out2, boxes2 = run()
            finally:
                C.cfg.TEST.BATCH_IMAGES = 1
            assert out2 == out1, nb
            for i, (a, b) in enumerate(zip(boxes1, boxes2)):
                assert np.array_equal(a, b), (nb, i)
    finally:
        C.cfg_set_mode("Test", old[0])
        C.cfg.EXP_DIR = old[1]
        C.cfg.TEST.NUM_PROPOSALS = old[2]
        shutil.rmtree(os.path.join(C.cfg.ROOT_DIR, "output", "harness_mixed"), ignore_errors=True)


def _run_tool(args, timeout=900, extra_env=None):
    env = dict(os.environ)
    env.update(extra_env or {})
    env["PYTHONPATH"] = os.pathsep.join([TOOLS] + ([env["PYTHONPATH"]] if env.get("PYTHONPATH") else []))
    return subprocess.run([sys.executable, os.path.join(TOOLS, args[0])] + args[1:], env=env, cwd=REPO,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)


def test_prop_az_cli_fresh_process(mods):
    """tools/prop_az.py as a fresh process (reference flags: tools/prop_az.py:30-54): full-size synthetic net,
    4 synthetic 600x1000 images, --tz; proposals.pkl layout / path; the same boxes as in-process im_propose."""
    torch, ffi, synth, HipAZNet, HipDetNet, orc = mods
    from detect import config as C
    exp = "cli_test_%d" % os.getpid()
    out_root = os.path.join(C.cfg.ROOT_DIR, "output", exp)
    try:
        r = _run_tool(["prop_az.py", "--gpu", "0", "--net", "synthetic", "--imdb", "synthetic_600x1000_4",
                       "--tz", "0.0", "--exp", exp, "--def", "ignored.prototxt", "--def_fc", "ignored_fc.prototxt"])
        assert r.returncode == 0, r.stdout[-3000:]
        out = r.stdout
        assert "Called with args:" in out and "Using config:" in out
        assert out.count("300 proposals, evaluate 739 regions, reaches depth 5.") == 4
        for i in range(1, 5):
            assert re.search(r"im_prop: %d/4 \d+\.\d{3}s" % i, out)
        assert "The recall is 0.000" in out and "The average proposal generation time is" in out
        pf = os.path.join(out_root, "synthetic_600x1000_4", "vgg16_az_net_synthetic_1234", "proposals.pkl")
        assert os.path.exists(pf), out[-2000:]
        with open(pf, "rb") as f:
            prop = pickle.load(f)
        assert sorted(prop.keys()) == ["boxes", "recall", "time"] and prop["recall"] == 0
        assert isinstance(prop["time"], float) and len(prop["boxes"]) == 4
        for b in prop["boxes"]:
            assert b.dtype == np.float64 and b.shape == (300, 4)
            assert b[:, 0].min() >= 0 and b[:, 2].max() <= 999 and b[:, 3].max() <= 599
        # the same net built in this process gives the same proposals (MIOpen may pick another conv algorithm
        # in another process, which moves conv5_3 by ulps: compare as sets with a pixel tolerance)
        sys.path.insert(0, TOOLS)
        import prop_az
        from detect import test as T
        old_tz = C.cfg.SEAR.get("Tz", 0.0)
        C.cfg_set_mode("Test", 0.0)
        net = prop_az.load_net("synthetic", 0)
        from datasets.factory import get_imdb
        imdb = get_imdb("synthetic_600x1000_4")
        for i in (0, 3):
            with redirect_stdout(io.StringIO()):
                Y = T.im_propose(net, imdb.image_at(i))
            hit = [np.abs(Y - r).max(axis=1).min() <= 5e-2 for r in prop["boxes"][i]]
            assert np.mean(hit) >= 0.95, (i, np.mean(hit))
        C.cfg_set_mode("Test", old_tz)
        # neither --tz nor --thresh: a usable message, not a TypeError
        r2 = _run_tool(["prop_az.py", "--net", "synthetic", "--imdb", "synthetic_600x1000_1"], timeout=120)
        assert r2.returncode == 2 and "--thresh / --tz is required" in r2.stdout
    finally:
        shutil.rmtree(out_root, ignore_errors=True)


def test_queued_image_pipeline_reads_its_own_image(mods):
    """detect.test's queue-ahead halves (_propose_start / _propose_finish: upload + front-end kernel + backbone + search of
    image i+1 enqueued while image i runs) against the synchronous sequence, with a backbone that really reads the blob (a
    narrow VGG16).  Regression: the front-end ran on the ctx stream when torch's current stream was the default stream
    (handle 0 = "ctx stream" to az_image_blob_dev_on), so the backbone could read a blob the front-end had not written yet --
    the previous image's, when the allocator handed the same block out again."""
    torch, ffi, synth, HipAZNet, HipDetNet, orc = mods
    from aznet_hip.backbone import VGG16Conv5
    from detect import config as C
    from detect import test as T
    old_tz = C.cfg.SEAR.get("Tz", 0.0)
    C.cfg_set_mode("Test", 0.0)
    try:
        bb = VGG16Conv5(device="cuda:0", seed=11, width_div=32)
        assert bb.out_channels == synth.SMALL_DIMS["C"]
        net = HipAZNet(synth.make_head(seed=77, **synth.SMALL_DIMS), backbone=bb, name="queued_pipeline")
        ims = [synth.make_scene_image(300 + j, 375, 500) for j in range(6)]
        bb.normalize_output(T._get_image_blob(ims[0], net)[0])
        want = []
        with redirect_stdout(io.StringIO()):
            for im in ims:
                blob, _ = T._get_image_blob(im, net)
                conv = net.compute_conv(blob)
                want.append((conv.clone(), T.im_propose(net, im)))
            pend, got = None, []
            order = list(range(len(ims))) * 6
            for k in range(len(order) + 1):
                nxt = T._propose_start(net, ims[order[k]], after=(pend["done"] if pend is not None else None)) if k < len(order) else None
                if pend is not None:
                    Y = T._propose_finish(net, pend)
                    got.append((pend["conv"], Y))
                pend = nxt
        for k, (conv, Y) in zip(order, got):
            assert torch.equal(conv, want[k][0]), "image %d: conv5_3 of the queued pipeline differs" % k
            assert np.array_equal(Y, want[k][1])
    finally:
        C.cfg_set_mode("Test", old_tz)


def _run_tool_ranks(world, args, timeout=900, extra_env=None):
    """tools/<args[0]> as `world` fresh ranks under torch.distributed.run on 127.0.0.1 (what the driver does for N > 1)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.update(extra_env or {})
    env["PYTHONPATH"] = os.pathsep.join([TOOLS] + ([env["PYTHONPATH"]] if env.get("PYTHONPATH") else []))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(TOOLS, args[0])] + args[1:]
    return subprocess.run(cmd, env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)


@pytest.mark.parametrize("variable_count", [False, True])
def test_prop_az_cli_two_ranks_equal_one(mods, tmp_path, variable_count):
    """The multi-rank branch of tools/prop_az.py (shard by image, queue-ahead, staged device records, ONE gather, rank 0
    writes proposals.pkl) as two fresh ranks sharing this box's GPU (--dist-backend gloo: the exchange over host tensors;
    RCCL needs a GPU per rank), a ragged image count (7: rank 0 owns 4 images, rank 1 owns 3 and sends a padding record):
    its proposals.pkl must hold, image by image, exactly the boxes the one-process run writes (test.py:492, 532-535: list
    order = image order).  variable_count: cfg.SEAR.FIXED_PROPOSAL_NUM off -- host records of a collectively agreed size."""
    torch, ffi, synth, HipAZNet, HipDetNet, orc = mods
    from detect import config as C
    base = "cli_ranks_%d_%d" % (os.getpid(), int(variable_count))
    # (fixed count: full trees, 300 proposals per image; variable count: a threshold that ends most trees early)
    common = ["--net", "synthetic", "--imdb", "synthetic_600x1000_7", "--tz", "0.35" if variable_count else "0.0",
              "--def", "x.prototxt", "--def_fc", "y.prototxt"]
    if variable_count:
        yml = tmp_path / "var.yml"
        yml.write_text("SEAR:\n  FIXED_PROPOSAL_NUM: False\n  Tc: 0.6\n")
        common += ["--cfg", str(yml)]
    roots = [os.path.join(C.cfg.ROOT_DIR, "output", base + s_) for s_ in ("_w1", "_w2")]
    try:
        r1 = _run_tool(["prop_az.py", "--gpu", "0", "--exp", base + "_w1"] + common)
        assert r1.returncode == 0, r1.stdout[-3000:]
        r2 = _run_tool_ranks(2, ["prop_az.py", "--exp", base + "_w2", "--dist-backend", "gloo"] + common)
        assert r2.returncode == 0, r2.stdout[-4000:]
        props = []
        for root in roots:
            pf = os.path.join(root, "synthetic_600x1000_7", "vgg16_az_net_synthetic_1234", "proposals.pkl")
            assert os.path.exists(pf), (r1.stdout[-1500:], r2.stdout[-1500:])
            with open(pf, "rb") as f:
                props.append(pickle.load(f))
        one, two = props
        assert sorted(two.keys()) == ["boxes", "recall", "time"] and two["recall"] == 0 and isinstance(two["time"], float)
        assert len(one["boxes"]) == len(two["boxes"]) == 7
        counts = set()
        for i, (a, b) in enumerate(zip(one["boxes"], two["boxes"])):
            assert a.dtype == b.dtype == np.float64 and a.shape == b.shape, (i, a.shape, b.shape)
            assert np.array_equal(a, b), "image %d: the two-rank run's boxes differ from the one-process run's" % i
            counts.add(a.shape[0])
        if variable_count:
            assert len(counts) > 1                            # (a data-dependent number of boxes per image)
        else:
            assert counts == {300}
        # every rank printed its own images' lines; rank 0 wrote the file
        assert r2.stdout.count("proposals, evaluate") == 7 and r2.stdout.count("wrote ") == 1
    finally:
        for root in roots:
            shutil.rmtree(root, ignore_errors=True)


def test_prop_az_cli_lockstep_batches_one_and_two_ranks(mods):
    """--batch-images 3 (cfg.TEST.BATCH_IMAGES): the one-process run and two fresh ranks (each searching ITS images in
    lockstep batches, the batch's records staged into the send buffer by one strided copy) write the proposals.pkl of the
    plain one-process run, box for box."""
    torch, ffi, synth, HipAZNet, HipDetNet, orc = mods
    from detect import config as C
    base = "cli_batches_%d" % os.getpid()
    common = ["--net", "synthetic", "--imdb", "synthetic_600x1000_7", "--tz", "0.35", "--def", "x.prototxt", "--def_fc", "y.prototxt"]
    roots = [os.path.join(C.cfg.ROOT_DIR, "output", base + s_) for s_ in ("_w1", "_w1b", "_w2b")]
    try:
        # (three PROCESSES compared box for box: the backbone's convolutions with the same summation order in each)
        det = {"AZ_BACKBONE_DETERMINISTIC": "1"}
        r1 = _run_tool(["prop_az.py", "--gpu", "0", "--exp", base + "_w1"] + common, extra_env=det)
        assert r1.returncode == 0, r1.stdout[-3000:]
        r1b = _run_tool(["prop_az.py", "--gpu", "0", "--exp", base + "_w1b", "--batch-images", "3"] + common, extra_env=det)
        assert r1b.returncode == 0, r1b.stdout[-3000:]
        r2b = _run_tool_ranks(2, ["prop_az.py", "--exp", base + "_w2b", "--dist-backend", "gloo", "--batch-images", "3"] + common,
                              extra_env=det)
        assert r2b.returncode == 0, r2b.stdout[-4000:]
        props = []
        for root in roots:
            pf = os.path.join(root, "synthetic_600x1000_7", "vgg16_az_net_synthetic_1234", "proposals.pkl")
            assert os.path.exists(pf), (r1b.stdout[-1500:], r2b.stdout[-1500:])
            with open(pf, "rb") as f:
                props.append(pickle.load(f))
        for other in props[1:]:
            assert len(other["boxes"]) == 7
            for i, (a, b) in enumerate(zip(props[0]["boxes"], other["boxes"])):
                assert a.dtype == b.dtype == np.float64 and np.array_equal(a, b), i
        assert r2b.stdout.count("proposals, evaluate") == 7 and r2b.stdout.count("wrote ") == 1
        # the per-image lines of the one-process runs agree (same counts, same order)
        lines = [[ln for ln in r.stdout.splitlines() if "proposals, evaluate" in ln] for r in (r1, r1b)]
        assert lines[0] == lines[1] and len(lines[0]) == 7
    finally:
        for root in roots:
            shutil.rmtree(root, ignore_errors=True)


# ---------------------------------------------------------------- .caffemodel row (f2)
def test_caffemodel_files_load_into_the_gpu_head(mods, tmp_path):
    """V1- and V2-format .caffemodel files (written with tests/test_caffemodel.py's protobuf encoder) read by
    aznet_hip.caffemodel, loaded through tools/prop_az.py:load_net, give the same head outputs -- bit for
    bit -- as the head loaded from memory; the conv layers land in the torch backbone unchanged."""
    torch, ffi, synth, HipAZNet, HipDetNet, orc = mods
    import test_caffemodel as tc
    sys.path.insert(0, TOOLS)
    import prop_az
    head = synth.make_head(seed=5, **synth.SMALL_DIMS)
    rng = np.random.RandomState(3)
    from aznet_hip.backbone import VGG16_CONV
    conv, cin = {}, 3
    for item in VGG16_CONV:
        if item == "P":
            continue
        name, cout = item
        cout = max(4, cout // 32)
        conv[name] = ((rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32),
                      (0.01 * rng.standard_normal(cout)).astype(np.float32))
        cin = cout
    assert cin == synth.SMALL_DIMS["C"]
    fmap = synth.make_feature_map(9, synth.SMALL_DIMS["C"], 38, 63)
    rois = np.array([[0, 0, 0, 999, 599], [0, 100, 50, 400, 300], [0, 8, 8, 8, 8], [0, 500, 300, 990, 590]], np.float32)
    ref_net = HipAZNet(head, name="mem")
    ref_net.set_conv(fmap)
    want = ref_net.ctx.head_forward(rois)
    blob = np.random.RandomState(1).standard_normal((1, 3, 64, 96)).astype(np.float32)
    for fmt in ("v1", "v2"):
        path = str(tmp_path / ("az_%s.caffemodel" % fmt))
        tc.write_az_caffemodel(path, head, conv, fmt)
        net = prop_az.load_net(path, 0)
        assert net.name == "az_%s" % fmt
        net.set_conv(fmap)
        got = net.ctx.head_forward(rois)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        # backbone weights arrived in Caffe layout [out, in, 3, 3] and are used as such
        for layer in net.backbone.layers:
            if layer is not None:
                assert np.array_equal(layer[1].cpu().numpy(), conv[layer[0]][0])
                assert np.array_equal(layer[2].cpu().numpy(), conv[layer[0]][1])
        from aznet_hip.backbone import VGG16Conv5
        mem = VGG16Conv5(device="cuda:0", weights=conv, width_div=32)
        assert torch.equal(net.backbone(blob), mem(blob))


# ---------------------------------------------------------------- backbone plumbing (g1)
def test_vgg16_conv5_plumbing(mods):
    """VGG16Conv5 (models/Pascal/VGG16/az-net/test.prototxt:16-384): 600x1000 -> [1,512,38,63] (four ceil-mode
    2x2/2 pools, no pool5), Caffe-layout weights used as given, and the GPU forward equals a CPU torch forward
    of the same stack within 1e-3."""
    torch, ffi, synth, HipAZNet, HipDetNet, orc = mods
    import torch.nn.functional as F
    from aznet_hip.backbone import VGG16Conv5, VGG16_CONV
    bb = VGG16Conv5(device="cuda:0", seed=11)
    names = [l[0] for l in bb.layers if l is not None]
    assert names == [x[0] for x in VGG16_CONV if x != "P"] and len(names) == 13 and bb.layers.count(None) == 4
    assert bb.layers[-1][0] == "conv5_3" and bb.out_channels == 512
    x = torch.from_numpy(np.random.RandomState(0).uniform(-120, 130, (1, 3, 600, 1000)).astype(np.float32))
    y = bb(x)
    assert tuple(y.shape) == (1, 512, 38, 63) and y.is_contiguous() and y.dtype == torch.float32
    assert (synth.conv_out_size(600), synth.conv_out_size(1000)) == (38, 63)
    assert tuple(bb(torch.zeros(1, 3, 375, 500)).shape) == (1, 512, 24, 32)       # odd sizes: ceil mode
    # CPU reference of the same graph
    h = x
    with torch.no_grad():
        for layer in bb.layers:
            if layer is None:
                h = F.max_pool2d(h, 2, 2, ceil_mode=True)
            else:
                h = F.relu(F.conv2d(h, layer[1].cpu(), layer[2].cpu(), padding=1))
    ref = h
    err = (y.cpu() - ref).abs().max().item()
    assert err <= 1e-3 * max(1.0, ref.abs().max().item()), err
    # weights round-trip: a dict of Caffe-layout arrays is what the next instance uses
    w = {l[0]: (l[1].cpu().numpy(), l[2].cpu().numpy()) for l in bb.layers if l is not None}
    bb2 = VGG16Conv5(device="cuda:0", weights=w)
    assert torch.equal(bb2(x), y)
    # and the map feeds the search: set_conv borrows the tensor (after synchronising torch's stream)
    net = HipAZNet(synth.make_head(seed=1, **synth.SMALL_DIMS), backbone=VGG16Conv5(device="cuda:0", seed=2, width_div=32),
                   name="plumb")
    conv = net.compute_conv(x)
    assert tuple(conv.shape) == (1, 16, 38, 63)
    Y = net.propose(ffi.AzContext.make_params(600, 1000, 1.0, 0.0))
    net2 = HipAZNet(synth.make_head(seed=1, **synth.SMALL_DIMS), name="plumb2")
    net2.set_conv(conv.cpu().numpy())
    assert np.array_equal(Y, net2.propose(ffi.AzContext.make_params(600, 1000, 1.0, 0.0)))


def test_backbone_epilogues_equal_pytorchs_ops_bit_for_bit(mods):
    """az_bias_relu / az_bias_relu_pool (az_epilogue.hip; test.prototxt:16-384: bias, in-place ReLU, MAX pool 2x2/2 in ceil
    mode) against PyTorch's own element-wise launches: same fp32 operations, so the same bits -- on even and odd map sizes, with
    non-zero biases and negative inputs -- and a channels_last VGG16Conv5 with the fused epilogues gives the map it gives
    without them (at fp32 tolerance: see below)."""
    torch, ffi, synth, HipAZNet, HipDetNet, orc = mods
    import torch.nn.functional as F
    g = torch.Generator(device="cpu").manual_seed(5)
    for C, H, W in [(64, 600, 1000), (128, 75, 125), (512, 38, 63), (8, 1, 1), (4, 3, 2), (256, 151, 7), (3, 5, 8), (7, 6, 12)]:
        y = (torch.randn(1, C, H, W, generator=g) * 3.0).cuda()
        b = torch.randn(C, generator=g).cuda()
        ref = F.relu(y + b.view(1, -1, 1, 1))
        refp = F.max_pool2d(ref, 2, 2, ceil_mode=True)
        if C % 4 == 0:                                           # channel-last: float4 over the channels
            ycl = y.clone().contiguous(memory_format=torch.channels_last)
            got = ffi.bias_relu_(ycl.clone(memory_format=torch.preserve_format), b)
            assert got.is_contiguous(memory_format=torch.channels_last) and torch.equal(got, ref)
            gp = ffi.bias_relu_pool(ycl, b)
            assert tuple(gp.shape) == tuple(refp.shape) and torch.equal(gp, refp)
        gotn = ffi.bias_relu_(y.clone(), b)                      # the NCHW form
        assert torch.equal(gotn, ref)
        gpn = ffi.bias_relu_pool(y, b)                           # the NCHW form
        assert gpn.is_contiguous() and torch.equal(gpn, refp)
    from aznet_hip.backbone import VGG16Conv5
    x = torch.from_numpy(np.random.RandomState(3).uniform(-120, 130, (1, 3, 375, 500)).astype(np.float32))
    bb = VGG16Conv5(device="cuda:0", seed=13, width_div=8, channels_last_compute=True, channels_last_out=True)
    for layer_i, layer in enumerate(bb.layers):                  # non-zero biases
        if layer is not None:
            bb.layers[layer_i] = (layer[0], layer[1], torch.randn(layer[2].shape, generator=g).cuda() * 0.1)
    # The whole stack, fused epilogues against PyTorch's own launches: MIOpen's convolutions do not repeat their bits from
    # call to call at every shape (two runs of the SAME PyTorch-only stack differ by an ulp from conv5_1 on: measured), so
    # the stack is compared at fp32 tolerance; the bit-for-bit statement is the per-launch one above.
    fused = bb(x)
    assert tuple(fused.shape) == (1, 64, 24, 32) and fused.is_contiguous(memory_format=torch.channels_last)
    bb.fused_epilogue = False
    plain = bb(x)
    assert torch.allclose(fused, plain, rtol=1e-5, atol=1e-5 * float(plain.abs().max()))
