"""HipAZNet: the object that stands where the reference's pair of caffe.Net objects stood
(tools/prop_az.py:92-98 builds `nets = {'full': net, 'fc': net_fc}`).

Two ways to use it, both without any CPU compute path:
  * whole search on the GPU -- `detect.test.im_propose(HipAZNet, im)` calls `propose()`,
    i.e. az_propose (the level loop never returns to the host);
  * as a pycaffe-shaped net -- `.blobs[name].reshape(...)` and
    `.forward(blobs=[...], data=... | conv5_3=..., rois=...)` return `zoom_prob`,
    `adj_prob`, `adj_bbox` (and `conv5_3` when asked), which is exactly the surface
    lib/detect/test.py:221-242 drives.  The reference's own Python loop (or the oracle's)
    can therefore run on top of the HIP head unchanged; tests use that to check the fused
    loop against the per-level path.
"""
import numpy as np

from aznet_hip import ffi


class _Blob(object):
    def __init__(self):
        self.shape = None

    def reshape(self, *shape):
        self.shape = tuple(shape)


class HipAZNet(object):
    def __init__(self, head, backbone=None, device=0, name="vgg16_az_net_hip", ctx=None,
                 max_regions=None, gemm_mode=None):
        self.ctx = ctx or ffi.AzContext(device, max_regions=max_regions, gemm_mode=gemm_mode)
        self.ctx.load_head(head)
        if ffi._default_ctx is None or getattr(ffi._default_ctx, "h", None) is None:
            ffi.set_default_context(self.ctx)      # drop-in helpers (nms, divide_region) use this rank's GPU
        self.backbone = backbone
        self.name = name
        self.blobs = {k: _Blob() for k in ("data", "rois", "conv5_3")}
        self._conv = None          # what the last forward/set_image left in HBM

    # dict-of-two compatibility: net['full'] / net['fc'] / 'fc' in net.keys()
    def __getitem__(self, k):
        if k in ("full", "fc"):
            return self
        raise KeyError(k)

    def keys(self):
        return ["full", "fc"]

    def __contains__(self, k):
        return k in ("full", "fc")

    # ---- feature map -----------------------------------------------------------------
    def set_conv(self, conv, wait=True):
        """conv: NumPy [1,C,H,W] (copied to HBM) or a CUDA torch tensor (borrowed; torch's current
        stream is synchronised before the ctx stream reads it).  wait=False: see
        AzContext.set_feature_map."""
        self.ctx.set_feature_map(conv, wait=wait)
        self._conv = conv

    def image_blob(self, im, pixel_means, scale):
        """_get_image_blob (lib/detect/test.py:27-59) on the GPU for a uint8 BGR image: with a
        backbone, the blob is written straight into a CUDA tensor the backbone reads (no f32 host
        copy); without one, a NumPy array comes back."""
        if self.backbone is None:
            return self.ctx.image_blob(im, pixel_means, scale)
        import torch
        oh, ow = self.ctx.image_blob_size(im.shape[0], im.shape[1], scale)
        out = torch.empty((1, 3, oh, ow), dtype=torch.float32, device=self.backbone.device)
        torch.cuda.current_stream(out.device).synchronize()
        return self.ctx.image_blob(im, pixel_means, scale, out=out)

    def image_blob_enqueue(self, im, pixel_means, scale):
        """As image_blob for a net with a backbone, without any host wait: upload and front-end kernel are enqueued on
        torch's current stream (where the backbone runs next), az_image_blob_dev_on."""
        import torch
        oh, ow = self.ctx.image_blob_size(im.shape[0], im.shape[1], scale)
        out = torch.empty((1, 3, oh, ow), dtype=torch.float32, device=self.backbone.device)
        return self.ctx.image_blob(im, pixel_means, scale, out=out,
                                   stream=torch.cuda.current_stream(out.device).cuda_stream)

    def compute_conv(self, data_blob):
        """Run the torch backbone on a [1,3,H,W] blob and hand conv5_3 to the HIP context."""
        if self.backbone is None:
            raise RuntimeError("HipAZNet has no backbone: supply conv5_3 with set_conv()")
        import torch
        conv = self.backbone(data_blob)
        self.set_conv(conv)                                    # (synchronises torch's stream first)
        return conv

    # ---- whole search ------------------------------------------------------------------
    def propose(self, params, want_scores=False, want_stats=False, stage=None):
        """stage (multi-GPU): a callable run between launch and fetch, e.g. DeviceGather.stage(j), which
        enqueues the device-to-device copy of the result record into the RCCL send buffer."""
        if stage is None:
            return self.ctx.propose(params, want_scores=want_scores, want_stats=want_stats)
        self.ctx.propose_launch(params)
        stage()
        return self.ctx.propose_fetch(want_scores=want_scores, want_stats=want_stats)

    # ---- pycaffe-shaped surface ----------------------------------------------------------
    def propose_batch(self, params, convs, want_scores=False, want_stats=False):
        """The images of consecutive iterations of the dataset loop (lib/detect/test.py:508-513), all of one shape, searched
        in lockstep (AzContext.batch_launch / batch_fetch): a list with every image's im_propose result."""
        self.ctx.batch_launch(params, convs)
        self._conv = convs[-1]
        return [self.ctx.batch_fetch(i, want_scores=want_scores, want_stats=want_stats) for i in range(len(convs))]

    def forward(self, blobs=None, **kw):
        rois = np.ascontiguousarray(kw["rois"], dtype=np.float32)
        if "conv5_3" in kw:
            conv = kw["conv5_3"]
            if conv is not self._conv:
                self.set_conv(conv)
        elif "data" in kw:
            self.compute_conv(kw["data"])
        z, p, d = self.ctx.head_forward(rois)
        out = {"zoom_prob": z, "adj_prob": p, "adj_bbox": d}
        if blobs:
            for b in blobs:
                out[b] = self._conv
        return out


class HipDetNet(object):
    """Fast R-CNN detection net on the shared conv map: stands where the reference's
    `frcnn_nets = {'fc': caffe.Net(frcnn/test_fc.prototxt, ...)}` stood (tools/test_shared.py).
    It shares the AZ net's az_ctx, so both heads read the same channel-last map in HBM.
    Also pycaffe-shaped (`forward(rois=, conv5_3=)` -> cls_prob, bbox_pred; test.py:302-307)."""

    def __init__(self, det_head, az_net, name="vgg16_frcnn_hip"):
        self.ctx = az_net.ctx
        self.az_net = az_net
        self.ctx.load_det_head(det_head)
        self.num_classes = self.ctx.det_dims["ncls"]
        self.name = name
        self.blobs = {k: _Blob() for k in ("data", "rois", "conv5_3")}

    def __getitem__(self, k):
        if k == "fc":
            return self
        raise KeyError(k)

    def keys(self):
        return ["fc"]

    def __contains__(self, k):
        return k == "fc"

    def detect(self, boxes, scale, im_shape, dedup, batch_size, eps):
        return self.ctx.detect(boxes, scale, im_shape[0], im_shape[1], dedup=dedup, batch_size=batch_size, eps=eps)

    def forward(self, blobs=None, **kw):
        rois = np.ascontiguousarray(kw["rois"], dtype=np.float32)
        if "conv5_3" in kw and kw["conv5_3"] is not self.az_net._conv:
            self.az_net.set_conv(kw["conv5_3"])
        p, b = self.ctx.det_forward(rois)
        return {"cls_prob": p, "bbox_pred": b}
