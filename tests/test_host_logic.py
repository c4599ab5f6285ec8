"""CPU: host-side mirror of the reference's config / driver interface (no GPU calls)."""
import os
import pickle

import numpy as np
import pytest


def test_config_defaults_and_yaml_merge(tmp_path):
    from detect import config as C
    cfg = C.cfg
    assert cfg.TEST.SCALES == (600,) and cfg.TEST.MAX_SIZE == 1000 and cfg.TEST.NUM_PROPOSALS == 300
    assert cfg.SEAR.MIN_SIDE == 10 and cfg.SEAR.BATCH_SIZE == 10000 and cfg.DEDUP_BOXES == 1. / 16.
    assert cfg.SEAR.NUM_SUBREG == 11 and cfg.EPS == 1e-14 and cfg.SEAR.FIXED_PROPOSAL_NUM is True
    y = tmp_path / "voc.yml"
    y.write_text("TEST:\n  MAX_SIZE: 800\nSEAR:\n  BATCH_SIZE: 1000\n  AZ_CONV: [conv5_3]\n")
    C.cfg_from_file(str(y))
    assert cfg.TEST.MAX_SIZE == 800 and cfg.SEAR.BATCH_SIZE == 1000
    bad = tmp_path / "bad.yml"
    bad.write_text("TEST:\n  NOPE: 1\n")
    with pytest.raises(KeyError):
        C.cfg_from_file(str(bad))
    bad.write_text("TEST:\n  MAX_SIZE: 'x'\n")
    with pytest.raises(ValueError):
        C.cfg_from_file(str(bad))
    C.cfg_set_mode("Test", 0.25)
    assert cfg.SEAR.Tz == 0.25 and cfg.SEAR.NUM_PROPOSALS == 300
    C.cfg_set_mode("Train")
    assert cfg.SEAR.Tz == 0.0 and cfg.SEAR.NUM_PROPOSALS == 2000
    with pytest.raises(AssertionError):
        C.cfg_set_mode("Test")
    t = tmp_path / "thresh.pkl"
    t.write_bytes(pickle.dumps(0.4321, protocol=2))
    assert C.cfg_load_thresh(str(t)) == 0.4321
    C.cfg_set_path(None)
    assert cfg.EXP_DIR == "default"
    C.cfg_set_path("exp1")

    class Imdb(object):
        name = "voc_2007_test"

    class Net(object):
        name = "net1"
    assert C.get_output_dir(Imdb(), Net()).endswith(os.path.join("output", "exp1", "voc_2007_test", "net1"))
    cfg.TEST.MAX_SIZE = 1000
    cfg.SEAR.BATCH_SIZE = 10000
    C.cfg_set_path(None)


def test_image_scale_and_blob():
    from detect import test as T
    from detect.config import cfg
    cfg.TEST.MAX_SIZE = 1000
    assert T._im_scale((600, 1000, 3)) == [1.0]
    assert T._im_scale((375, 500, 3)) == [1.6]
    assert abs(T._im_scale((500, 1000, 3))[0] - 1.0) < 1e-12          # capped by MAX_SIZE
    # the blob itself is made by a HIP kernel (tests/test_gpu_parity.py); here: its input contract
    assert T._as_uint8(np.full((4, 5, 3), 128.0)).dtype == np.uint8
    with pytest.raises(TypeError):
        T._as_uint8(np.full((4, 5, 3), 0.5))


def test_synthetic_imdb_and_timer():
    from datasets.factory import get_imdb
    from utils.timer import Timer
    db = get_imdb("synthetic_600x1000_3")
    assert len(db.image_index) == 3 and db.image_at(1).shape == (600, 1000, 3) and db.name == "synthetic_600x1000_3"
    assert np.array_equal(db.image_at(2), db.image_at(2))
    with pytest.raises(IOError):
        get_imdb("voc_2007_test")                 # known name, but no VOCdevkit offline
    with pytest.raises(KeyError):
        get_imdb("coco_2014_val")
    roidb = db.roidb
    assert len(roidb) == 3 and roidb[0]["boxes"].shape[1] == 4 and (roidb[0]["gt_classes"] > 0).all()
    t = Timer()
    t.tic()
    assert t.toc() >= 0 and t.calls == 1


def test_ffi_struct_layout_matches_header():
    """az_params / az_stats in ffi.py must mirror include/aznet_hip.h field for field."""
    import ctypes
    from aznet_hip import ffi
    assert ctypes.sizeof(ffi.AzParams) == 8 + 6 * 8 + 4 * 4
    # + spec_rows, root_deferred, static_plan, n_passes, pass_rows[16], search_form, n_reruns, pass_levels[16]
    assert ctypes.sizeof(ffi.AzStats) == 5 * 4 + 3 * 16 * 4 + 3 * 4 + 4 + 16 * 4 + 2 * 4 + 16 * 4
    hdr0 = open(os.path.join(os.path.dirname(__file__), "..", "include", "aznet_hip.h")).read()
    st_body = hdr0[hdr0.index("typedef struct {", hdr0.index("} az_params;")):hdr0.index("} az_stats;")]
    import re as _re
    names = _re.findall(r"int32_t\s+(\w+)(?:\[\w+\])?;", st_body)
    assert names == [f[0] for f in ffi.AzStats._fields_], (names, [f[0] for f in ffi.AzStats._fields_])
    p = ffi.AzContext.make_params(600, 1000, 1.0, 0.3, num_proposals=300, batch_size=1000, speculate=False)
    assert (p.im_h, p.im_w, p.Tz, p.batch_size, p.fixed_num, p.reserved) == (600, 1000, 0.3, 1000, 1, 1)
    # the flag bits of params.reserved (include/aznet_hip.h)
    mk = lambda **kw: ffi.AzContext.make_params(600, 1000, 1.0, 0.3, **kw).reserved      # noqa: E731
    assert mk() == 0
    assert mk(fused=False) == 2 and mk(tune=True) == 4 and mk(radix_select=True) == 8 and mk(fused_levels=False) == 16
    assert mk(static_tree=False) == 32
    assert mk(pair_spec=False) == 64 and mk(pair_spec=True) == 128
    assert mk(full_spec=False) == 256 and mk(full_spec=True) == 512
    assert mk(pair_spec=False, full_spec=True, static_tree=False) == 64 + 512 + 32
    assert mk(full_spec="closure") == 512 + 1024
    import re
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "aznet_hip.h")).read()
    for bit in range(11):
        assert re.search(r"bit %d\b" % bit, hdr), "params.reserved bit %d is not documented in the header" % bit


def _make_voc_tree(root, year="2007"):
    """A two-image VOCdevkit<year> tree: ImageSets/Main/test.txt, Annotations/*.xml, JPEGImages/*.jpg."""
    from PIL import Image
    base = os.path.join(root, "VOCdevkit" + year, "VOC" + year)
    for d in ("ImageSets/Main", "Annotations", "JPEGImages"):
        os.makedirs(os.path.join(base, d))
    with open(os.path.join(base, "ImageSets", "Main", "test.txt"), "w") as f:
        f.write("000001\n000002\n\n")
    objs = {"000001": [("dog", 48, 240, 195, 371), ("person", 8, 12, 352, 498)],
            "000002": [("Train ", 139, 200, 207, 301)]}
    rng = np.random.RandomState(0)
    for idx, lst in objs.items():
        xml = "<annotation><filename>%s.jpg</filename>" % idx
        for name, x1, y1, x2, y2 in lst:
            xml += ("<object><name>%s</name><difficult>0</difficult><bndbox><xmin>%d</xmin><ymin>%d</ymin>"
                    "<xmax>%d</xmax><ymax>%d</ymax></bndbox></object>" % (name, x1, y1, x2, y2))
        xml += "</annotation>"
        with open(os.path.join(base, "Annotations", idx + ".xml"), "w") as f:
            f.write(xml)
        Image.fromarray(rng.randint(0, 256, (60, 80, 3)).astype(np.uint8)).save(
            os.path.join(base, "JPEGImages", idx + ".jpg"), quality=95)
    return os.path.join(root, "VOCdevkit" + year)


def test_pascal_voc_reader(tmp_path):
    """Index file, image paths, 0-based uint16 boxes and class ids as lib/datasets/pascal_voc.py:50-143."""
    import datasets
    from datasets.pascal_voc import pascal_voc, VOC_CLASSES
    devkit = _make_voc_tree(str(tmp_path))
    db = pascal_voc("test", "2007", devkit_path=devkit)
    assert db.name == "voc_2007_test" and db.num_classes == 21 and db.image_index == ["000001", "000002"]
    assert db.image_path_at(1).endswith(os.path.join("JPEGImages", "000002.jpg"))
    roidb = db.gt_roidb(use_cache=False)
    assert roidb[0]["boxes"].dtype == np.uint16
    assert roidb[0]["boxes"].tolist() == [[47, 239, 194, 370], [7, 11, 351, 497]]
    assert roidb[0]["gt_classes"].tolist() == [VOC_CLASSES.index("dog"), VOC_CLASSES.index("person")]
    assert roidb[1]["gt_classes"].tolist() == [VOC_CLASSES.index("train")]       # lower().strip()
    im = db.image_at(0)
    assert im.shape == (60, 80, 3) and im.dtype == np.uint8
    from PIL import Image
    rgb = np.asarray(Image.open(db.image_path_at(0)).convert("RGB"))
    assert np.array_equal(im, rgb[:, :, ::-1])                                    # BGR, as cv2.imread
    with pytest.raises(IOError):
        pascal_voc("val", "2007", devkit_path=devkit)                             # no such image set file
    assert "voc_2007_test" in datasets.factory.list_imdbs()


def test_evaluate_recall_host_math():
    """imdb.evaluate_recall around the matching: skipped images, ignored classes, curve and AR
    (lib/datasets/imdb.py:120-159), with the oracle standing in for the GPU matching call."""
    from datasets.imdb import imdb
    from oracle import az_oracle as orc
    from helpers import load
    g = load("g10_recall.npz")
    n = int(g["n_img"])

    class Db(imdb):
        def gt_roidb(self):
            return [{"boxes": g["gt%d" % i], "gt_classes": g["cls%d" % i]} for i in range(n)]

    class FakeCtx(object):
        def recall_match(self, cands, gts):
            return orc.recall_gt_overlaps(cands, gts)

    db = Db("golden")
    db._image_index = list(range(n))
    ar, gt_overlaps, recalls, thresholds = db.evaluate_recall([g["cand%d" % i] for i in range(n)], ctx=FakeCtx())
    assert ar == float(g["ar"]) and np.array_equal(gt_overlaps, g["gt_overlaps"])
    assert np.array_equal(recalls, g["recalls"]) and np.array_equal(thresholds, g["thresholds"])


def test_bench_self_launch_relays_rank_failures():
    """`python bench.py --gpus 2` must start its own ranks (torch.distributed.run on 127.0.0.1) without touching
    the GPU in the parent, and exit non-zero when a rank fails.  On this CPU-only box every rank fails at
    device selection, which is exactly the case to relay: no JSON line, non-zero exit code."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-e2e"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=300)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert r.returncode == 0 and '"n_gpus": 2' in r.stdout
    else:
        assert r.returncode != 0
        assert '"metric"' not in r.stdout
        assert "torch.distributed" in r.stderr or "ChildFailedError" in r.stderr or "Error" in r.stderr


def test_bench_line_is_compact_strict_json():
    """The driver keeps 8 KB of stdout: bench.py's ONE printed line must be strict JSON well under that, carry the contract's
    keys plus `roofline` and `cpu_baseline`, and stay small whatever the side legs put into the full record (round 5's
    30 KB line came back `parsed: null`).  Run on round 5's full record and on one with hostile values."""
    import json
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import bench
    full = json.load(open(os.path.join(repo, "profiles", "r05_bench_default.json")))
    assert len(json.dumps(full)) > 8192                      # the record that broke the driver
    for rec in (full, dict(full, junk={"x": ["y" * 100] * 1000}, value=float(np.float32(2.5e5)),
                          roofline=dict(full["roofline"], frac=float("nan"), note="n" * 20000),
                          config=dict(full["config"], workload="w" * 50000, gather="g" * 5000),
                          cpu_baseline=dict(full["cpu_baseline"], sample="s" * 9000))):
        line = bench.compact_line(rec, extras_file="bench_extras.json")
        assert "\n" not in line and len(line) < 4096 < 8192
        got = json.loads(line, parse_constant=lambda c: pytest.fail("non-strict JSON constant %s" % c))
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "gpu_over_cpu", "rccl"):
            assert k in got, k
        assert "workload" in got["config"] and "model" not in got["config"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
            assert k in got["roofline"], k
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in got["cpu_baseline"], k
        assert got["rccl"]["world"] == got["n_gpus"]
        assert got["vs_baseline"] is None and got["unit"] == "proposals/s"
    # the default run has no side legs: they are opt-in
    src = open(os.path.join(repo, "bench.py")).read()
    assert '"--extras"' in src and "write_extras(" in src


def test_batched_proposals_groups_by_shape_and_keeps_dataset_order(monkeypatch, capsys):
    """detect.test._batched_proposals (cfg.TEST.BATCH_IMAGES): every image exactly once and in dataset order, a batch = at most
    nb images of any shapes and level counts, taken from a read-ahead window -- only images too small for the lockstep form
    (fewer than three levels) are kept among themselves --, the per-image line printed in order; with the GPU halves replaced
    by recorders."""
    from detect import test as T
    shapes = [(375, 500, 3), (50, 70, 3), (375, 500, 3), (600, 1000, 3), (500, 375, 3), (60, 60, 3), (375, 500, 3),
              (333, 500, 3), (90, 120, 3), (375, 500, 3), (160, 200, 3), (375, 500, 3), (375, 500, 3), (500, 375, 3),
              (375, 500, 3), (45, 220, 3), (375, 500, 3)]
    levels = [T._lockstep_ok(sh) for sh in shapes]
    assert T._num_levels((600, 1000, 3)) == 6 and T._num_levels((333, 500, 3)) == 6 and T._num_levels((90, 120, 3)) == 4
    assert levels.count(False) == 3 and not T._lockstep_ok((79, 500, 3)) and T._lockstep_ok((80, 500, 3))
    ims = [np.full(s, i, dtype=np.uint8) for i, s in enumerate(shapes)]
    batches, log = [], []

    def backbones(net, group, after=None):
        assert len({T._lockstep_ok(im.shape) for im in group}) == 1
        h = {"shapes": [im.shape for im in group], "n": len(group), "ims": group, "convs": [("conv", int(im.flat[0])) for im in group],
             "after": after}
        log.append(("backbones", [int(im.flat[0]) for im in group]))
        return h

    def launch(net, h):
        h["done"] = ("done", tuple(int(im.flat[0]) for im in h["ims"]))
        batches.append([int(im.flat[0]) for im in h["ims"]])
        log.append(("launch", batches[-1]))
        return h

    def finish(net, h, i, quiet=False):
        k = int(h["ims"][i].flat[0])
        return np.full((2, 4), float(k)), "line %d" % k
    monkeypatch.setattr(T, "_batch_backbones", backbones)
    monkeypatch.setattr(T, "_batch_launch", launch)
    monkeypatch.setattr(T, "_batch_finish", finish)
    for nb, ahead in ((4, True), (4, False), (2, True), (16, True), (1, True)):
        del batches[:], log[:]
        out = list(T._batched_proposals(None, iter(ims), len(ims), nb, launch_ahead=ahead))
        printed = capsys.readouterr().out.split("\n")[:-1]
        assert [int(im.flat[0]) for im, _, _ in out] == list(range(len(ims)))           # dataset order, each once
        assert printed == ["line %d" % i for i in range(len(ims))]
        assert all(float(Y[0, 0]) == i for i, (_, Y, _) in enumerate(out))
        assert all(conv[T.cfg.SEAR.FRCNN_CONV[0]] == ("conv", i) for i, (_, _, conv) in enumerate(out))
        assert sorted(i for b in batches for i in b) == list(range(len(ims)))
        assert all(1 <= len(b) <= nb and len({levels[i] for i in b}) == 1 and b == sorted(b) for b in batches)
        if nb >= 4:
            assert any(len({shapes[i] for i in b}) > 1 for b in batches)                 # shapes mix inside a batch
        # a batch starts with the oldest unprocessed image and reaches at most a window of 4 nb images ahead
        seen = set()
        for b in batches:
            assert b[0] == min(set(range(len(ims))) - seen)
            assert max(b) < b[0] + 4 * nb + len(b)
            seen.update(b)
        if nb >= 4:
            assert any(len(b) > 1 and b != list(range(b[0], b[0] + len(b))) for b in batches)   # images were pulled forward
        # launch_ahead: the next batch's search is enqueued before the current batch's images are handed out
        launches = [i for i, e in enumerate(log) if e[0] == "launch"]
        assert len(launches) == len(batches)
