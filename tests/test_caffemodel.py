"""CPU: the .caffemodel reader (protobuf wire format, no Caffe) against a real protobuf encoder
and against its own minimal writer."""
import numpy as np
import pytest

from aznet_hip import caffemodel as cm
from aznet_hip import synth


def _proto_classes():
    """caffe.proto's weight-bearing subset, built at run time with the protobuf package."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="mini_caffe.proto", package="mini", syntax="proto2")

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for fname, num, ftype, label, tname, packed in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                f.type_name = ".mini." + tname
            if packed:
                f.options.packed = True
    REP, OPT = F.LABEL_REPEATED, F.LABEL_OPTIONAL
    msg("BlobShape", [("dim", 1, F.TYPE_INT64, REP, None, True)])
    msg("BlobProto", [("shape", 7, F.TYPE_MESSAGE, OPT, "BlobShape", False),
                      ("data", 5, F.TYPE_FLOAT, REP, None, True),
                      ("num", 1, F.TYPE_INT32, OPT, None, False), ("channels", 2, F.TYPE_INT32, OPT, None, False),
                      ("height", 3, F.TYPE_INT32, OPT, None, False), ("width", 4, F.TYPE_INT32, OPT, None, False)])
    msg("LayerParameter", [("name", 1, F.TYPE_STRING, OPT, None, False), ("type", 2, F.TYPE_STRING, OPT, None, False),
                           ("bottom", 3, F.TYPE_STRING, REP, None, False),
                           ("blobs", 7, F.TYPE_MESSAGE, REP, "BlobProto", False)])
    msg("V1LayerParameter", [("name", 4, F.TYPE_STRING, OPT, None, False), ("type", 5, F.TYPE_INT32, OPT, None, False),
                             ("blobs", 6, F.TYPE_MESSAGE, REP, "BlobProto", False)])
    msg("NetParameter", [("name", 1, F.TYPE_STRING, OPT, None, False),
                         ("layers", 2, F.TYPE_MESSAGE, REP, "V1LayerParameter", False),
                         ("input", 3, F.TYPE_STRING, REP, None, False),
                         ("layer", 100, F.TYPE_MESSAGE, REP, "LayerParameter", False)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("mini.NetParameter"))


def write_az_caffemodel(path, head, conv, fmt):
    """An AZ-Net .caffemodel (InnerProduct head + VGG16 conv layers) serialised by the REAL protobuf encoder:
    fmt 'v2' = NetParameter.layer (LayerParameter, BlobShape dims), 'v1' = NetParameter.layers
    (V1LayerParameter, legacy num/channels/height/width).  Also used by tests/test_gpu_harness.py."""
    Net = _proto_classes()
    net = Net(name="az_" + fmt)
    net.input.append("data")
    named = [("int6", head["W6"], head["b6"]), ("int7_1", head["W71"], head["b71"]), ("int7_2", head["W72"], head["b72"]),
             ("adj_score", head["Was"], head["bas"]), ("adj_bbox", head["Wab"], head["bab"]),
             ("zoom_score", head["Wz"], head["bz"])]
    named = [(k, w, b) for k, (w, b) in conv.items()] + named

    def put(blobs, a):
        bl = blobs.add()
        if fmt == "v2":
            bl.shape.dim.extend(a.shape)
        else:
            sh = (1,) * (4 - a.ndim) + tuple(a.shape)
            bl.num, bl.channels, bl.height, bl.width = [int(x) for x in sh]
        bl.data.extend(np.asarray(a, dtype=np.float32).ravel().tolist())
    for name, w, b in named:
        if fmt == "v2":
            lay = net.layer.add(name=name, type="Convolution" if name.startswith("conv") else "InnerProduct")
            net.layer.add(name="relu_" + name, type="ReLU")           # weightless layers in between
        else:
            lay = net.layers.add(name=name, type=4 if name.startswith("conv") else 14)
        put(lay.blobs, w)
        put(lay.blobs, b)
    with open(path, "wb") as f:
        f.write(net.SerializeToString())


def test_reader_against_real_protobuf_encoder(tmp_path):
    Net = _proto_classes()
    rng = np.random.RandomState(0)
    w = rng.randn(6, 10).astype(np.float32)
    b = rng.randn(6).astype(np.float32)
    cw = rng.randn(4, 3, 3, 3).astype(np.float32)
    net = Net(name="n")
    net.input.append("data")
    lay = net.layer.add(name="int6", type="InnerProduct")
    lay.bottom.append("pool5")
    for a in (w, b):
        bl = lay.blobs.add()
        bl.shape.dim.extend(a.shape)
        bl.data.extend(a.ravel().tolist())
    net.layer.add(name="relu6", type="ReLU")                       # no blobs: skipped
    v1 = net.layers.add(name="conv1_1", type=4)                   # V1 layer with legacy 4-D shape
    bl = v1.blobs.add(num=4, channels=3, height=3, width=3)
    bl.data.extend(cw.ravel().tolist())
    bl = v1.blobs.add(num=1, channels=1, height=1, width=4)
    bl.data.extend([1, 2, 3, 4])
    f = tmp_path / "x.caffemodel"
    f.write_bytes(net.SerializeToString())
    layers = cm.load_caffemodel(str(f))
    assert sorted(layers) == ["conv1_1", "int6"]
    assert np.array_equal(layers["int6"][0], w) and np.array_equal(layers["int6"][1], b)
    assert layers["conv1_1"][0].shape == (4, 3, 3, 3) and np.array_equal(layers["conv1_1"][0], cw)
    assert layers["conv1_1"][1].shape == (1, 1, 1, 4)
    bb = cm.backbone_from_layers(layers)
    assert bb["conv1_1"][0].shape == (4, 3, 3, 3) and bb["conv1_1"][1].shape == (4,)


@pytest.mark.parametrize("v1,legacy", [(False, False), (False, True), (True, True)])
def test_heads_roundtrip_through_writer(tmp_path, v1, legacy):
    head = synth.make_head(seed=5, **synth.SMALL_DIMS)
    det = synth.make_det_head(seed=6, **synth.SMALL_DET_DIMS)
    layers = {"int6": [head["W6"], head["b6"]], "int7_1": [head["W71"], head["b71"]],
              "int7_2": [head["W72"], head["b72"]], "adj_score": [head["Was"], head["bas"]],
              "adj_bbox": [head["Wab"], head["bab"]], "zoom_score": [head["Wz"], head["bz"]]}
    f = tmp_path / "az.caffemodel"
    cm.write_caffemodel(str(f), layers, v1=v1, legacy_shapes=legacy)
    got = cm.az_head_from_layers(cm.load_caffemodel(str(f)))
    for k in head:
        assert got[k].shape == head[k].shape and np.array_equal(got[k], head[k]), k
    dl = {"fc6": [det["W6"], det["b6"]], "fc7": [det["W7"], det["b7"]], "cls_score": [det["Wc"], det["bc"]],
          "bbox_pred": [det["Wb"], det["bb"]]}
    f2 = tmp_path / "det.caffemodel"
    cm.write_caffemodel(str(f2), dl, v1=v1, legacy_shapes=legacy)
    got = cm.det_head_from_layers(cm.load_caffemodel(str(f2)))
    for k in det:
        assert np.array_equal(got[k], det[k]), k
    with pytest.raises(KeyError):
        cm.az_head_from_layers(cm.load_caffemodel(str(f2)))


@pytest.mark.parametrize("fmt", ["v1", "v2"])
def test_full_net_files_from_the_real_encoder(tmp_path, fmt):
    """Head + conv layers written by the protobuf package in both layer formats -> identical arrays back."""
    head = synth.make_head(seed=5, **synth.SMALL_DIMS)
    rng = np.random.RandomState(2)
    conv = {"conv1_1": (rng.randn(4, 3, 3, 3).astype(np.float32), rng.randn(4).astype(np.float32)),
            "conv5_3": (rng.randn(16, 4, 3, 3).astype(np.float32), rng.randn(16).astype(np.float32))}
    f = tmp_path / "net.caffemodel"
    write_az_caffemodel(str(f), head, conv, fmt)
    layers = cm.load_caffemodel(str(f))
    got = cm.az_head_from_layers(layers)
    for k in head:
        assert got[k].shape == head[k].shape and np.array_equal(got[k], head[k]), k
    bb = cm.backbone_from_layers(layers)
    for k, (w, b) in conv.items():
        assert np.array_equal(bb[k][0], w) and np.array_equal(bb[k][1], b)
