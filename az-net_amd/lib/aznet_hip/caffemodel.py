"""Read the weights of a `.caffemodel` without Caffe or a compiled caffe.proto.

A .caffemodel is a serialized `NetParameter` protobuf message.  Only the few fields that hold
layer names and weight blobs are needed, so this module walks the protobuf WIRE FORMAT
directly (varint / length-delimited / fixed32 / fixed64 records) with the field numbers of the
public caffe.proto:

    NetParameter        layer = 100 (LayerParameter, current format)
                        layers = 2  (V1LayerParameter, the 2014-2015 format)
    LayerParameter      name = 1, type = 2 (string), blobs = 7
    V1LayerParameter    name = 4, type = 5 (enum),   blobs = 6
    BlobProto           shape = 7 (BlobShape), data = 5 (packed float), double_data = 8,
                        legacy 4-D shape: num = 1, channels = 2, height = 3, width = 4
    BlobShape           dim = 1 (packed int64)

It stands in for `caffe.Net(prototxt, caffemodel, caffe.TEST)` loading the parameters
(tools/prop_az.py:92-96, tools/test_shared.py): the layer graphs are fixed in this
implementation, so only the blobs matter.  `write_caffemodel` is the matching minimal encoder,
used by the tests to build fixtures.
"""
import numpy as np


# ---- wire-format primitives ----------------------------------------------------------------
def _varint(buf, pos):
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not (b & 0x80):
            return result, pos
        shift += 7


def _fields(buf, start, end):
    """Yield (field_number, wire_type, value_or_span) over buf[start:end]; length-delimited
    fields yield a (start, end) span so that big blobs are never copied while walking."""
    pos = start
    while pos < end:
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
            yield fn, wt, v
        elif wt == 1:
            yield fn, wt, (pos, pos + 8)
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            yield fn, wt, (pos, pos + n)
            pos += n
        elif wt == 5:
            yield fn, wt, (pos, pos + 4)
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d at byte %d" % (wt, pos))


def _packed_ints(buf, span):
    out = []
    pos, end = span
    while pos < end:
        v, pos = _varint(buf, pos)
        out.append(v)
    return out


def _blob(buf, span):
    """BlobProto -> float32 ndarray with its shape."""
    legacy = {}
    shape = None
    chunks = []
    dchunks = []
    for fn, wt, v in _fields(buf, *span):
        if fn == 7 and wt == 2:                       # BlobShape
            for f2, w2, v2 in _fields(buf, *v):
                if f2 == 1:
                    shape = (shape or []) + (_packed_ints(buf, v2) if w2 == 2 else [v2])
        elif fn == 5:                                 # data (packed, or one float per record)
            chunks.append(np.frombuffer(buf, dtype="<f4", count=(v[1] - v[0]) // 4, offset=v[0]))
        elif fn == 8:                                 # double_data
            dchunks.append(np.frombuffer(buf, dtype="<f8", count=(v[1] - v[0]) // 8, offset=v[0]))
        elif fn in (1, 2, 3, 4) and wt == 0:
            legacy[fn] = v
    if chunks:
        data = np.concatenate(chunks) if len(chunks) > 1 else chunks[0]
    elif dchunks:
        data = np.concatenate(dchunks).astype(np.float32)
    else:
        data = np.zeros(0, dtype=np.float32)
    if shape is None:
        shape = [legacy.get(i, 1) for i in (1, 2, 3, 4)] if legacy else [data.size]
    return np.array(data, dtype=np.float32).reshape([int(d) for d in shape])


def _layer(buf, span, name_field, blobs_field, type_field):
    name, ltype, blobs = None, None, []
    for fn, wt, v in _fields(buf, *span):
        if fn == name_field and wt == 2:
            name = bytes(buf[v[0]:v[1]]).decode("utf-8", "replace")
        elif fn == blobs_field and wt == 2:
            blobs.append(_blob(buf, v))
        elif fn == type_field:
            ltype = bytes(buf[v[0]:v[1]]).decode("utf-8", "replace") if wt == 2 else int(v)
    return name, ltype, blobs


def load_caffemodel(path):
    """{layer name: [ndarray, ...]} for every layer that carries blobs (both layer formats)."""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    layers = {}
    for fn, wt, v in _fields(buf, 0, len(buf)):
        if wt != 2:
            continue
        if fn == 100:
            name, _, blobs = _layer(buf, v, 1, 7, 2)
        elif fn == 2:
            name, _, blobs = _layer(buf, v, 4, 6, 5)
        else:
            continue
        if name is not None and blobs:
            layers[name] = blobs
    return layers


# ---- mapping onto this implementation's heads ---------------------------------------------------
def _fc(layers, name):
    """InnerProduct blob as [out, in] (legacy models store it as [1, 1, out, in])."""
    w, b = layers[name][0], layers[name][1]
    b = np.ascontiguousarray(b.ravel(), dtype=np.float32)
    w = np.ascontiguousarray(w.reshape(b.size, -1), dtype=np.float32)
    return w, b


def az_head_from_layers(layers):
    """AZ head of models/*/VGG16/az-net/test_fc.prototxt: int6, int7_1, int7_2, adj_score,
    adj_bbox, zoom_score -> the dict az_load_head / synth.make_head use."""
    W6, b6 = _fc(layers, "int6")
    W71, b71 = _fc(layers, "int7_1")
    W72, b72 = _fc(layers, "int7_2")
    Was, bas = _fc(layers, "adj_score")
    Wab, bab = _fc(layers, "adj_bbox")
    Wz, bz = _fc(layers, "zoom_score")
    return {"W6": W6, "b6": b6, "W71": W71, "b71": b71, "W72": W72, "b72": b72, "Was": Was, "bas": bas,
            "Wab": Wab, "bab": bab, "Wz": Wz, "bz": bz}


def det_head_from_layers(layers):
    """Fast R-CNN head of models/*/VGG16/frcnn/test_fc.prototxt: fc6, fc7, cls_score, bbox_pred."""
    W6, b6 = _fc(layers, "fc6")
    W7, b7 = _fc(layers, "fc7")
    Wc, bc = _fc(layers, "cls_score")
    Wb, bb = _fc(layers, "bbox_pred")
    return {"W6": W6, "b6": b6, "W7": W7, "b7": b7, "Wc": Wc, "bc": bc, "Wb": Wb, "bb": bb}


def backbone_from_layers(layers):
    """{conv name: (W [out, in, 3, 3], b [out])} for conv1_1 .. conv5_3 (VGG16Conv5(weights=...))."""
    out = {}
    for name, blobs in layers.items():
        if name.startswith("conv") and len(blobs) >= 2 and blobs[0].ndim == 4:
            out[name] = (np.ascontiguousarray(blobs[0], dtype=np.float32),
                         np.ascontiguousarray(blobs[1].ravel(), dtype=np.float32))
    return out


# ---- minimal encoder (fixtures) ------------------------------------------------------------------
def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _enc_ld(fn, payload):
    return _enc_varint((fn << 3) | 2) + _enc_varint(len(payload)) + payload


def _enc_blob(a, legacy):
    a = np.ascontiguousarray(a, dtype="<f4")
    if legacy:
        dims = ([1] * (4 - a.ndim) + list(a.shape))[-4:]
        head = b"".join(_enc_varint((i << 3) | 0) + _enc_varint(int(d)) for i, d in zip((1, 2, 3, 4), dims))
    else:
        head = _enc_ld(7, _enc_ld(1, b"".join(_enc_varint(int(d)) for d in a.shape)))
    return head + _enc_ld(5, a.tobytes())


def write_caffemodel(path, layers, v1=False, legacy_shapes=False):
    """layers: {name: [ndarray, ...]}.  v1 selects the V1LayerParameter format (field 2)."""
    out = bytearray(_enc_ld(1, b"synthetic"))
    for name, blobs in layers.items():
        if v1:
            body = _enc_ld(4, name.encode()) + b"".join(_enc_ld(6, _enc_blob(b, legacy_shapes or True)) for b in blobs)
            out += _enc_ld(2, body)
        else:
            body = _enc_ld(1, name.encode()) + _enc_ld(2, b"InnerProduct") + \
                b"".join(_enc_ld(7, _enc_blob(b, legacy_shapes)) for b in blobs)
            out += _enc_ld(100, body)
    with open(path, "wb") as f:
        f.write(bytes(out))


__all__ = ["load_caffemodel", "az_head_from_layers", "det_head_from_layers", "backbone_from_layers",
           "write_caffemodel"]
