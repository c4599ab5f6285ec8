"""Seeded synthetic inputs for the AZ proposal path (no datasets or .caffemodel
files exist offline).  Pure NumPy; shared by tests, bench.py and the golden
generator.  Layouts are Caffe's: InnerProduct weights are row-major [out, in]
(models/Pascal/VGG16/az-net/test_fc.prototxt:26-220)."""
import numpy as np

# Layer sizes of the AZ head (test_fc.prototxt): roi_pool5 C*7*7 -> int6 4096 ->
# {int7_1 1024 -> adj_score 11 / adj_bbox 44 ; int7_2 256 -> zoom_score 1}.
FULL_DIMS = dict(C=512, n6=4096, n71=1024, n72=256)
SMALL_DIMS = dict(C=16, n6=128, n71=64, n72=32)
NUM_SUBREG = 11


def make_head(seed=1234, C=512, n6=4096, n71=1024, n72=256, pooled=7):
    """Random fp32 head weights scaled so zoom / adjacency scores spread over (0,1)
    without saturating (ties in the top-K sort would make index parity ill-defined)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    K6 = C * pooled * pooled

    def w(n_out, n_in, gain):
        a = rng.standard_normal((n_out, n_in), dtype=np.float32)
        a *= np.float32(gain / np.sqrt(n_in))
        return a

    head = {
        "W6": w(n6, K6, 1.0), "b6": (0.1 * rng.standard_normal(n6, dtype=np.float32)),
        "W71": w(n71, n6, 1.4), "b71": (0.1 * rng.standard_normal(n71, dtype=np.float32)),
        "W72": w(n72, n6, 1.4), "b72": (0.1 * rng.standard_normal(n72, dtype=np.float32)),
        "Was": w(NUM_SUBREG, n71, 2.0), "bas": np.zeros(NUM_SUBREG, dtype=np.float32),
        "Wab": w(4 * NUM_SUBREG, n71, 0.35), "bab": np.zeros(4 * NUM_SUBREG, dtype=np.float32),
        "Wz": w(1, n72, 2.0), "bz": np.zeros(1, dtype=np.float32),
    }
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in head.items()}


def make_feature_map(seed, C, H, W):
    """A post-ReLU-like conv5_3 stand-in: [1, C, H, W] f32, ~50 % zeros."""
    rng = np.random.Generator(np.random.PCG64(10_000 + seed))
    a = rng.standard_normal((1, C, H, W), dtype=np.float32)
    np.maximum(a, 0, out=a)
    return a


def make_image(seed, height=600, width=1000):
    """uint8 BGR image, `RandomState(seed).randint(0, 256)`."""
    return np.random.RandomState(seed).randint(0, 256, size=(height, width, 3)).astype(np.uint8)


def make_scene_image(seed, height=600, width=1000):
    """uint8 BGR image with STRUCTURE that differs from seed to seed: a flat or softly shaded background carrying 0..7
    textured rectangles ("objects") of random size, contrast and position.  make_image's uniform noise gives every image the
    same statistics everywhere, hence trees of the same density at a given Tz; a dataset does not -- some images are nearly
    empty, some are crowded, and the zoom tree follows the content.  Used where a STREAM of distinct images is the point
    (bench.py: stream_tz, tests/test_gpu_stream.py)."""
    rng = np.random.RandomState(50_000 + seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    base = rng.uniform(40, 200, 3).astype(np.float32)
    gx, gy = rng.uniform(-0.08, 0.08, 2)
    im = base[None, None, :] + (gx * (xx - width / 2) + gy * (yy - height / 2))[:, :, None]
    im += rng.normal(0, rng.uniform(1, 12), (height, width, 1)).astype(np.float32)
    n_obj = int(rng.choice([0, 0, 1, 1, 2, 3, 4, 5, 7]))
    for _ in range(n_obj):
        w = int(rng.uniform(0.04, 0.6) * width)
        h = int(rng.uniform(0.04, 0.6) * height)
        x0 = int(rng.uniform(0, width - w))
        y0 = int(rng.uniform(0, height - h))
        col = rng.uniform(0, 255, 3).astype(np.float32)
        period = rng.uniform(3, 40)
        ang = rng.uniform(0, np.pi)
        tex = np.sin((np.cos(ang) * xx[y0:y0 + h, x0:x0 + w] + np.sin(ang) * yy[y0:y0 + h, x0:x0 + w]) * (2 * np.pi / period))
        amp = rng.uniform(10, 110)
        im[y0:y0 + h, x0:x0 + w, :] = col[None, None, :] + amp * tex[:, :, None] + \
            rng.normal(0, rng.uniform(0, 25), (h, w, 3)).astype(np.float32)
    return np.clip(np.rint(im), 0, 255).astype(np.uint8)


def make_scene_map(seed, C, H, W):
    """A conv5_3 stand-in whose content differs from seed to seed the way make_scene_image's does (tests without a backbone):
    post-ReLU noise whose gain varies over 0..6 rectangular patches on a weak background, so that some maps drive deep trees
    and others end after a level or two at the same Tz."""
    rng = np.random.Generator(np.random.PCG64(70_000 + seed))
    a = rng.standard_normal((1, C, H, W), dtype=np.float32)
    gain = np.full((H, W), rng.uniform(0.15, 1.0), dtype=np.float32)
    for _ in range(int(rng.integers(0, 7))):
        h = int(rng.integers(2, max(3, H // 2)))
        w = int(rng.integers(2, max(3, W // 2)))
        y0 = int(rng.integers(0, H - h + 1))
        x0 = int(rng.integers(0, W - w + 1))
        gain[y0:y0 + h, x0:x0 + w] = rng.uniform(0.3, 2.5)
    a *= gain[None, None, :, :]
    a += rng.uniform(-0.6, 0.3)
    np.maximum(a, 0, out=a)
    return a


def make_object_head(seed=1234, zoom_channel=0, gamma=10.0, beta=0.12, delta=3.0, noise=0.1, clip=0.5, **dims):
    """make_head with a planted path through the zoom branch, so that the zoom indicator behaves the way a TRAINED AZ-Net's
    does -- high for a region that contains an object small against the region, falling as the region shrinks onto the
    object, low where there is none; hence consistent along an object's path down the tree -- instead of drifting with region
    size as random weights make it:
        s               = sum over the 49 bins of roi_pool5[zoom_channel]      (how much of the window the object fills)
        int6 units 0, 1 = relu(s), relu(s - clip)                              (W6 rows 0 and 1, b6[1] = -clip)
        int7_2 unit 0   = int6[0] - int6[1] = min(s, clip)                     ("there is an object in the window")
        int7_2 unit 1   = int6[0] = s
        zoom_score      = gamma * int7_2[0] - beta * int7_2[1] - delta + noise * (the random rest)       (Wz, bz)
    With maps from make_object_map (zoom_channel is zero except on a few planted blobs, each >= clip) the zoom tree at a
    tuned Tz follows the planted objects down the levels until a region is about the object's size: deep and sparse, a
    handful of regions per level, different from image to image.  Everything else (adjacency scores and boxes, the other
    units) is make_head's.  Caffe layout: W6 column c * 49 + p."""
    head = make_head(seed=seed, **dims)
    n6, K6 = head["W6"].shape
    C = K6 // 49
    assert 0 <= zoom_channel < C and n6 >= 2 and head["W72"].shape[0] >= 2
    for j, b in ((0, 0.0), (1, -clip)):
        head["W6"][j, :] = 0.0
        head["W6"][j, zoom_channel * 49:(zoom_channel + 1) * 49] = 1.0
        head["b6"][j] = b
    head["W71"][:, 0:2] = 0.0                    # (the planted units feed the zoom branch only)
    head["W72"][:, 0:2] = 0.0
    head["W72"][0:2, :] = 0.0
    head["W72"][0, 0], head["W72"][0, 1] = 1.0, -1.0
    head["W72"][1, 0] = 1.0
    head["b72"][0:2] = 0.0
    head["Wz"] *= np.float32(noise)
    head["Wz"][0, 0], head["Wz"][0, 1] = gamma, -beta
    head["bz"][0] = -delta
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in head.items()}


def make_object_map(seed, C, H, W, zoom_channel=0, max_objects=4):
    """A conv5_3 stand-in for make_object_head: make_feature_map's noise in every channel but `zoom_channel`, which is zero
    except on 0..max_objects planted blobs of 1x1..3x3 cells (16..48 px objects) with amplitudes in [0.6, 2]."""
    a = make_feature_map(20_000 + seed, C, H, W)
    rng = np.random.Generator(np.random.PCG64(90_000 + seed))
    a[0, zoom_channel] = 0.0
    for _ in range(int(rng.integers(0, max_objects + 1))):
        h, w = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        y0, x0 = int(rng.integers(0, max(1, H - h + 1))), int(rng.integers(0, max(1, W - w + 1)))
        a[0, zoom_channel, y0:y0 + h, x0:x0 + w] = np.float32(rng.uniform(0.6, 2.0))
    return a


def conv_out_size(n):
    """Spatial size after VGG16's four ceil-mode 2x2/2 max-pools
    (models/Pascal/VGG16/az-net/test.prototxt:16-384)."""
    for _ in range(4):
        n = (n + 1) // 2
    return n


# Fast R-CNN head (models/Pascal/VGG16/frcnn/test_fc.prototxt): roi_pool5 -> fc6 -> fc7 ->
# {cls_score ncls (softmax), bbox_pred 4*ncls}; VOC has 21 classes.
FULL_DET_DIMS = dict(C=512, n6=4096, n7=4096, ncls=21)
SMALL_DET_DIMS = dict(C=16, n6=128, n7=96, ncls=21)


def make_det_head(seed=4242, C=512, n6=4096, n7=4096, ncls=21, pooled=7):
    """Random fp32 detection-head weights (Caffe [out, in] layout)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    K6 = C * pooled * pooled

    def w(n_out, n_in, gain):
        a = rng.standard_normal((n_out, n_in), dtype=np.float32)
        a *= np.float32(gain / np.sqrt(n_in))
        return a

    head = {
        "W6": w(n6, K6, 1.0), "b6": 0.1 * rng.standard_normal(n6, dtype=np.float32),
        "W7": w(n7, n6, 1.4), "b7": 0.1 * rng.standard_normal(n7, dtype=np.float32),
        "Wc": w(ncls, n7, 3.0), "bc": np.zeros(ncls, dtype=np.float32),
        "Wb": w(4 * ncls, n7, 0.3), "bb": np.zeros(4 * ncls, dtype=np.float32),
    }
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in head.items()}
