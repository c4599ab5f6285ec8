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
