"""ctypes binding of libaznet_hip.so (include/aznet_hip.h).

This is the only way the Python host code reaches the GPU path, and there is no
fallback: if the shared library is missing or no gfx950 device is visible, the
constructors raise.  NumPy owns every host buffer; torch (when used) owns the
feature map and hands over a raw device pointer.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libaznet_hip.so")

AZ_MAX_LEVELS = 16
AZ_NUM_SUBREG = 11
AZ_OK = 0
AZ_BATCH_MAX = 32          # include/aznet_hip.h
AZ_ERR_INVALID, AZ_ERR_HIP, AZ_ERR_CAPACITY, AZ_ERR_STATE, AZ_ERR_NO_DEVICE = -1, -2, -3, -4, -5
_ERR_NAMES = {-1: "AZ_ERR_INVALID", -2: "AZ_ERR_HIP", -3: "AZ_ERR_CAPACITY", -4: "AZ_ERR_STATE",
              -5: "AZ_ERR_NO_DEVICE"}

# every symbol include/aznet_hip.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "az_version", "az_create", "az_destroy", "az_last_error", "az_set_limits", "az_load_head",
    "az_set_feature_map_dev", "az_set_feature_map_host", "az_propose", "az_propose_launch",
    "az_propose_fetch", "az_last_candidates", "az_divide_region", "az_sift_dup", "az_roi_dedup",
    "az_roi_pool", "az_head_forward", "az_decode_filter", "az_topk", "az_nms", "az_set_profiling",
    "az_last_kernel_times", "az_stream", "az_load_det_head", "az_det_forward", "az_detect",
    "az_set_gemm_mode", "az_last_anchors", "az_tune_begin", "az_tune_end", "az_tune_kth_largest",
    "az_tune_top", "az_tune_push", "az_bbox_overlaps", "az_recall_match", "az_image_blob_size",
    "az_image_blob_host", "az_image_blob_dev", "az_nms_batched", "az_set_graphs",
    "az_set_feature_map_dev_async", "az_result_record_layout", "az_propose_stage_result_dev",
    "az_propose_launch_on", "az_set_feature_map_dev_nhwc", "az_set_pass_costs", "az_get_pass_costs",
    "az_measure_box", "az_image_blob_dev_on", "az_set_lanes", "az_next_stream", "az_last_stream",
    "az_rccl_unique_id", "az_rccl_init", "az_gather_records", "az_rccl_destroy", "az_comm_stream",
    "az_bias_relu", "az_bias_relu_pool", "az_batch_launch", "az_batch_fetch", "az_batch_next_stream",
    "az_batch_stage_results_dev", "az_batch_fetch_all", "az_batch_launch_shapes", "az_abi_sizes",
]


class AzParams(ctypes.Structure):
    _fields_ = [("im_h", ctypes.c_int32), ("im_w", ctypes.c_int32), ("scale", ctypes.c_double),
                ("Tz", ctypes.c_double), ("Tc", ctypes.c_double), ("dedup", ctypes.c_double),
                ("eps", ctypes.c_double), ("min_side", ctypes.c_double),
                ("batch_size", ctypes.c_int32), ("num_proposals", ctypes.c_int32),
                ("fixed_num", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class AzStats(ctypes.Structure):
    _fields_ = [("n_proposals", ctypes.c_int32), ("num_eval", ctypes.c_int32),
                ("depth", ctypes.c_int32), ("n_levels", ctypes.c_int32),
                ("n_candidates", ctypes.c_int32),
                ("level_regions", ctypes.c_int32 * AZ_MAX_LEVELS),
                ("level_unique", ctypes.c_int32 * AZ_MAX_LEVELS),
                ("level_zoomed", ctypes.c_int32 * AZ_MAX_LEVELS),
                ("spec_rows", ctypes.c_int32), ("root_deferred", ctypes.c_int32),
                ("static_plan", ctypes.c_int32), ("n_passes", ctypes.c_int32),
                ("pass_rows", ctypes.c_int32 * AZ_MAX_LEVELS),
                ("search_form", ctypes.c_int32), ("n_reruns", ctypes.c_int32),
                ("pass_levels", ctypes.c_int32 * AZ_MAX_LEVELS)]


SEARCH_FORMS = {0: "level_loop", 1: "pair_speculation", 2: "whole_tree_pass", 3: "closure_pass", 4: "one_pass_plan",
                5: "batch_level_loop"}


class AzError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "%s (%d): %s" % (_ERR_NAMES.get(code, "AZ_ERR"), code, msg))
        self.code = code


_lib = None


def load_library(path=None):
    """Load libaznet_hip.so and declare prototypes.  Fails loudly when it is absent:
    build it with `python -c 'import __graft_entry__ as g; g.build()'` or
    `make -C az-net_amd/csrc`."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("AZNET_HIP_LIB") or LIB_PATH      # (AZNET_HIP_LIB: an A/B build of the library, measurements)
    if not os.path.exists(p):
        raise ImportError("libaznet_hip.so not found at %s -- the HIP extension is required "
                          "(no CPU fallback); run make -C az-net_amd/csrc" % p)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 and must be the
    # first to load it -- if this library pulls in /opt/rocm's copy first, torch later finds
    # "No HIP GPUs".  The host side uses torch for device memory / streams anyway.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = ctypes.CDLL(p)
    vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    fp, dp = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)
    ip, i64p = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)
    cip = ctypes.POINTER(ctypes.c_int)
    L.az_version.restype = ctypes.c_char_p
    L.az_version.argtypes = []
    L.az_create.argtypes = [ci, ctypes.POINTER(vp)]
    L.az_destroy.argtypes = [vp]
    L.az_last_error.restype = ctypes.c_char_p
    L.az_last_error.argtypes = [vp]
    L.az_set_limits.argtypes = [vp, ci, ci]
    L.az_set_gemm_mode.argtypes = [vp, ci]
    L.az_load_head.argtypes = [vp, ci, ci, ci, ci] + [fp] * 12
    L.az_set_feature_map_dev.argtypes = [vp, vp, ci, ci, ci]
    L.az_set_feature_map_host.argtypes = [vp, fp, ci, ci, ci]
    L.az_set_feature_map_dev_async.argtypes = [vp, vp, ci, ci, ci]
    szp = ctypes.POINTER(ctypes.c_size_t)
    L.az_result_record_layout.argtypes = [ci, szp, szp, szp, szp]
    L.az_propose_stage_result_dev.argtypes = [vp, vp, ctypes.c_size_t]
    L.az_propose.argtypes = [vp, ctypes.POINTER(AzParams), dp, fp, ci, cip, ctypes.POINTER(AzStats)]
    L.az_propose_launch.argtypes = [vp, ctypes.POINTER(AzParams)]
    L.az_propose_launch_on.argtypes = [vp, ctypes.POINTER(AzParams), vp, ci, ci, ci, ci]
    L.az_set_feature_map_dev_nhwc.argtypes = [vp, vp, ci, ci, ci]
    L.az_propose_fetch.argtypes = [vp, dp, fp, ci, cip, ctypes.POINTER(AzStats)]
    L.az_last_candidates.argtypes = [vp, dp, fp, ci, cip]
    L.az_divide_region.argtypes = [vp, dp, ci, cd, dp, ci, cip]
    L.az_sift_dup.argtypes = [vp, dp, ci, cd, dp, ci, cip]
    L.az_roi_dedup.argtypes = [vp, dp, ci, cd, cd, ci, fp, ip, ip, cip]
    L.az_roi_pool.argtypes = [vp, fp, ci, fp]
    L.az_head_forward.argtypes = [vp, fp, ci, fp, fp, fp]
    L.az_decode_filter.argtypes = [vp, dp, fp, fp, ci, ci, ci, cd, cd, dp, fp, ci, cip]
    L.az_topk.argtypes = [vp, fp, ci, ci, ip, cip]
    L.az_nms.argtypes = [vp, fp, ci, cd, i64p, cip]
    L.az_nms_batched.argtypes = [vp, fp, ip, ci, cd, i64p, ip]
    L.az_load_det_head.argtypes = [vp, ci, ci, ci, ci] + [fp] * 8
    L.az_det_forward.argtypes = [vp, fp, ci, fp, fp]
    L.az_detect.argtypes = [vp, dp, ci, cd, cd, ci, ci, ci, cd, fp, dp]
    L.az_set_profiling.argtypes = [vp, ci]
    L.az_set_graphs.argtypes = [vp, ci]
    L.az_last_kernel_times.argtypes = [vp, ctypes.c_char_p, fp, ip, ci, cip]
    L.az_stream.restype = vp
    L.az_stream.argtypes = [vp]
    L.az_set_pass_costs.argtypes = [vp, ci, ip, dp]
    L.az_get_pass_costs.argtypes = [vp, ip, dp, ci, cip]
    L.az_measure_box.argtypes = [vp, dp, dp]
    L.az_set_lanes.argtypes = [vp, ci]
    L.az_rccl_unique_id.argtypes = [vp, ctypes.c_size_t]
    L.az_rccl_init.argtypes = [vp, vp, ctypes.c_size_t, ci, ci]
    L.az_gather_records.argtypes = [vp, vp, vp, ctypes.c_size_t]
    L.az_rccl_destroy.argtypes = [vp]
    L.az_comm_stream.restype = vp
    L.az_comm_stream.argtypes = [vp]
    L.az_next_stream.restype = vp
    L.az_next_stream.argtypes = [vp]
    L.az_last_stream.restype = vp
    L.az_last_stream.argtypes = [vp]
    L.az_batch_next_stream.restype = vp
    L.az_batch_next_stream.argtypes = [vp]
    L.az_batch_launch.argtypes = [vp, ci, ctypes.POINTER(AzParams), ctypes.POINTER(vp), ci, ci, ci]
    L.az_batch_fetch.argtypes = [vp, ci, dp, fp, ci, cip, ctypes.POINTER(AzStats)]
    L.az_batch_launch_shapes.argtypes = [vp, ci, ctypes.POINTER(AzParams), ctypes.POINTER(vp), ci, cip, cip]
    L.az_batch_stage_results_dev.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_size_t]
    L.az_batch_fetch_all.argtypes = [vp, dp, fp, ci, cip, ctypes.POINTER(AzStats)]
    ll, llp = ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)
    u8p = ctypes.POINTER(ctypes.c_uint8)
    L.az_last_anchors.argtypes = [vp, dp, fp, ci, cip]
    L.az_tune_begin.argtypes = [vp, ll]
    L.az_tune_end.argtypes = [vp]
    L.az_tune_kth_largest.argtypes = [vp, ll, fp, llp]
    L.az_tune_top.argtypes = [vp, ll, fp, ll, llp]
    L.az_tune_push.argtypes = [vp, fp, ll]
    L.az_bbox_overlaps.argtypes = [vp, dp, ci, dp, ci, dp]
    L.az_recall_match.argtypes = [vp, ci, dp, ip, dp, ip, dp]
    L.az_image_blob_size.argtypes = [ci, ci, cd, cip, cip]
    L.az_image_blob_host.argtypes = [vp, u8p, ci, ci, fp, cd, fp, ci, ci]
    L.az_image_blob_dev.argtypes = [vp, u8p, ci, ci, fp, cd, vp, ci, ci]
    L.az_image_blob_dev_on.argtypes = [vp, u8p, ci, ci, fp, cd, vp, ci, ci, vp]
    L.az_bias_relu.argtypes = [vp, vp, vp, ci, ctypes.c_longlong, ci]
    L.az_bias_relu_pool.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci]
    for name in SYMBOLS:
        if name not in ("az_version", "az_last_error", "az_stream", "az_next_stream", "az_last_stream", "az_comm_stream",
                        "az_batch_next_stream"):
            getattr(L, name).restype = ci
    # the structs this module hands across the boundary have the library's layout (every fetch clears sizeof(az_stats) bytes
    # of the caller's block: a binding built for another header must not get that far)
    L.az_abi_sizes.argtypes = []
    sizes = int(L.az_abi_sizes())
    if (sizes & 0xffff, (sizes >> 16) & 0xffff) != (ctypes.sizeof(AzParams), ctypes.sizeof(AzStats)):
        raise AzError(AZ_ERR_INVALID, "libaznet_hip.so %s has sizeof(az_params, az_stats) = (%d, %d), this binding (%d, %d): "
                      "rebuild az-net_amd/csrc" % (L.az_version().decode(), sizes & 0xffff, (sizes >> 16) & 0xffff,
                                                  ctypes.sizeof(AzParams), ctypes.sizeof(AzStats)))
    if path is None:
        _lib = L
    return L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


class AzContext(object):
    """One GPU's search context (az_ctx).  Not thread-safe; one per process/GPU."""

    def __init__(self, device=0, max_regions=None, max_candidates=None, gemm_mode=None):
        """gemm_mode: 0 fp32 MFMA (default); 2 = int6 on the 16-bit matrix cores with fp32 operands as two
        fp16 terms (3 MFMAs per product, ~2^-21); 3 = as three bf16 terms (6 MFMAs, all 24 mantissa bits)
        (az_set_gemm_mode); None reads the AZ_GEMM_MODE environment variable (default 0)."""
        self.L = load_library()
        h = ctypes.c_void_p()
        rc = self.L.az_create(int(device), ctypes.byref(h))
        if rc != AZ_OK:
            raise AzError(rc, "az_create(device=%d) failed: a gfx950 GPU is required, there is no "
                              "CPU fallback" % device)
        self.h = h
        self.device = int(device)
        self.dims = None
        self.feat_shape = None
        self._feat_keepalive = None
        if max_regions is not None:
            self._chk(self.L.az_set_limits(self.h, int(max_regions),
                                           int(max_candidates or max_regions * AZ_NUM_SUBREG)))
        if gemm_mode is None:
            gemm_mode = int(os.environ.get("AZ_GEMM_MODE", "0"))
        if int(gemm_mode) not in (0, 2, 3):
            self.close()
            raise ValueError("gemm_mode must be 0 (fp32 MFMA), 2 (two fp16 terms) or 3 (three bf16 terms), got %r" % (gemm_mode,))
        self.gemm_mode = int(gemm_mode)
        if self.gemm_mode:
            self._chk(self.L.az_set_gemm_mode(self.h, self.gemm_mode))
        self.max_regions = max_regions or 16384
        self.max_candidates = max_candidates or self.max_regions * AZ_NUM_SUBREG

    def _chk(self, rc):
        if rc != AZ_OK:
            raise AzError(rc, self.L.az_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.az_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- setup --------------------------------------------------------------------
    def load_head(self, head):
        """head: dict of Caffe-layout fp32 arrays W6,b6,W71,b71,W72,b72,Was,bas,Wab,bab,Wz,bz."""
        W6 = _f32(head["W6"])
        n6, K6 = W6.shape
        assert K6 % 49 == 0
        C = K6 // 49
        n71, n72 = head["W71"].shape[0], head["W72"].shape[0]
        assert head["W71"].shape == (n71, n6) and head["W72"].shape == (n72, n6)
        assert head["Was"].shape == (11, n71) and head["Wab"].shape == (44, n71)
        assert head["Wz"].shape == (1, n72)
        arrs = [W6] + [_f32(head[k]) for k in ("b6", "W71", "b71", "W72", "b72", "Was", "bas", "Wab",
                                                 "bab", "Wz", "bz")]
        self._chk(self.L.az_load_head(self.h, C, n6, n71, n72, *[_p(a, ctypes.c_float) for a in arrs]))
        self.dims = dict(C=C, n6=n6, n71=n71, n72=n72, K6=K6)

    def set_feature_map(self, fmap, wait=True, producer_done=False):
        """fmap: [1,C,H,W] or [C,H,W]; a NumPy array (copied to HBM) or a CUDA torch tensor
        (borrowed: its data_ptr is handed to the library, the tensor is kept alive here).
        The ctx stream is not torch's: torch's current stream is synchronised first, so a map the
        backbone is still writing is never read early.  wait=False (torch tensors only) skips the
        closing synchronisation of the ctx stream (az_set_feature_map_dev_async): the tensor then
        has to stay untouched until the next propose/propose_fetch returns.  producer_done=True (torch tensors): the caller
        knows the map is complete (e.g. a search that read it has been fetched): torch's stream is not synchronised."""
        if isinstance(fmap, np.ndarray):
            a = _f32(fmap)
            if a.ndim == 4:
                assert a.shape[0] == 1
                a = a[0]
            C, H, W = a.shape
            self._chk(self.L.az_set_feature_map_host(self.h, _p(a, ctypes.c_float), C, H, W))
            self._feat_keepalive = None
        else:   # torch tensor on this device
            t, cl = self._torch_map(fmap)
            C, H, W = (int(x) for x in t.shape)
            import torch
            if not producer_done:
                torch.cuda.current_stream(t.device).synchronize()     # producer (backbone) done
            fn = self.L.az_set_feature_map_dev_nhwc if cl else (
                self.L.az_set_feature_map_dev if wait else self.L.az_set_feature_map_dev_async)
            self._chk(fn(self.h, ctypes.c_void_p(t.data_ptr()), C, H, W))
            self._feat_keepalive = fmap
        self.feat_shape = (C, H, W)

    def _torch_map(self, fmap):
        """(tensor [C,H,W] view, channels_last?) of a CUDA conv5_3 tensor: NCHW-contiguous maps are transposed into
        ctx memory by the library, torch.channels_last ones ([1,C,H,W] stored [H][W][C]) are borrowed as they are."""
        import torch
        t = fmap
        assert t.is_cuda and str(t.dtype) == "torch.float32" and t.device.index == self.device, \
            "feature map must be a float32 CUDA tensor on this context's GPU"
        if t.dim() == 4:
            assert t.shape[0] == 1
            if t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous():
                return t[0], True
            t = t[0]
        assert t.is_contiguous()
        return t, False

    # ---- hot path -----------------------------------------------------------------
    @staticmethod
    def make_params(im_h, im_w, scale, Tz, num_proposals=300, fixed_num=True, Tc=0.05,
                    dedup=1. / 16., eps=1e-14, min_side=10, batch_size=10000, speculate=True, fused=True,
                    tune=False, radix_select=False, fused_levels=True, static_tree=True, pair_spec=None, full_spec=None,
                    early_end=True):
        """speculate=False evaluates levels 1-3 one by one instead of in one pass (same bits,
        slower); fused=False keeps the geometry of those levels as separate launches;
        fused_levels=False does the same for the levels after them (az_level.hip).  All
        exist for tests and measurements.  tune=True selects the tuner's variant of the search
        (lib/detect/tune.py:256-316) and keeps the anchor history (last_anchors).  radix_select=True
        does the final top-k with the single-workgroup radix select (same result, for tests).
        static_tree=False: with Tz <= 0 (every zoom test passes, the tree depends on the image shape only) still
        walk the tree level by level instead of forwarding all levels' rois in one head pass (same bits).
        pair_spec: None = let the context decide from its previous search whether a level's head pass also carries the
        rows of ALL children of its regions (so that the next level needs no pass); False = never; True = at every
        eligible level (same bits in all three).
        full_spec: None = let the context decide from its previous search of this image shape whether the search's ONE
        head pass evaluates the rows of the shape's full tree, every level finding its outputs by RoIPool window (pays
        for dense trees: after a full tree the full tree's rows, otherwise the closure rows); False = never; True = the
        full tree's rows whenever the shape allows; "closure" = the closure rows whenever the shape allows -- every region
        any pruning of the tree can produce, so no Tz can miss a window (same bits in all four).
        early_end=False: enqueue every level even when the context's previous search of the shape ended early (by default
        such a search is enqueued only up to the level where that one ended, and run again in full if this tree goes on;
        same bits)."""
        return AzParams(int(im_h), int(im_w), float(scale), float(Tz), float(Tc), float(dedup),
                        float(eps), float(min_side), int(batch_size), int(num_proposals),
                        1 if fixed_num else 0,
                        (0 if speculate else 1) | (0 if fused else 2) | (4 if tune else 0) |
                        (8 if radix_select else 0) | (0 if fused_levels else 16) | (0 if static_tree else 32) |
                        (0 if pair_spec is None else (128 if pair_spec else 64)) |
                        (0 if full_spec is None else ((512 | 1024) if full_spec == "closure" else (512 if full_spec else 256))) |
                        (0 if early_end else 4096))

    def propose(self, params, want_scores=False, want_stats=False):
        cap = params.num_proposals if params.fixed_num else self.max_candidates
        boxes = np.empty((cap, 4), dtype=np.float64)
        scores = np.empty((cap,), dtype=np.float32)
        n = ctypes.c_int(0)
        st = AzStats()
        self._chk(self.L.az_propose(self.h, ctypes.byref(params), _p(boxes, ctypes.c_double),
                                    _p(scores, ctypes.c_float), cap, ctypes.byref(n), ctypes.byref(st)))
        out = [boxes[:n.value].copy() if n.value < cap else boxes]
        if want_scores:
            out.append(scores[:n.value].copy())
        if want_stats:
            out.append(st)
        return out[0] if len(out) == 1 else tuple(out)

    def _ext(self, handle):
        """torch view of one of the context's HIP streams (by raw handle)."""
        import torch
        cache = self.__dict__.setdefault("_ext_streams", {})
        s = cache.get(int(handle))
        if s is None:
            s = cache[int(handle)] = torch.cuda.ExternalStream(int(handle), device=torch.device("cuda", self.device))
        return s

    def wait_event(self, event, params=None):
        """Make the stream the NEXT launched search runs on wait (on the device, no host synchronisation) for a
        torch.cuda.Event -- e.g. the one recorded behind the backbone's last kernel on torch's stream.  params: the
        parameters that search will be launched with -- searches that cannot be queued (a data-dependent proposal count, the
        tuner's variant) always run on the context's first lane, whatever lane is next in turn (az_capi.hip: next_lane)."""
        first_lane = params is not None and (not params.fixed_num or (params.reserved & 4))
        h = self.L.az_stream(self.h) if first_lane else self.L.az_next_stream(self.h)
        self._ext(h).wait_event(event)

    def set_lanes(self, lanes):
        """2: queued searches take turns between two streams of this context, so consecutive images overlap on the GPU
        (az_set_lanes); 1: one stream (the default)."""
        self._chk(self.L.az_set_lanes(self.h, int(lanes)))
        self.lanes = int(lanes)

    def record_event(self):
        """A torch.cuda.Event recorded now on the stream of the search launched last: behind that search."""
        return self._ext(self.L.az_last_stream(self.h)).record_event()

    def propose_launch(self, params, fmap=None, producer_done=False, producer_event=None):
        """fmap (a CUDA torch tensor [1,C,H,W] / [C,H,W] on this GPU): hand the image's map over in the same call
        (az_propose_launch_on); it must stay untouched until propose_fetch returns.  producer_done=True skips the
        synchronisation of torch's current stream (the caller knows the map is complete); producer_event (a
        torch.cuda.Event recorded behind the map's producer) orders the search behind it ON THE DEVICE instead, so the
        host can go on and enqueue the next image's backbone while this search runs.
        Up to three searches per lane may be launched before the first is fetched (fixed proposal count): the host then enqueues
        the next image's launch sequence while the GPU still works on the current one -- same stream, the searches do not
        overlap on the GPU; propose_fetch returns them oldest first."""
        self._last_params = params
        if not hasattr(self, "_queued"):
            self._queued = []
        if fmap is None:
            self._chk(self.L.az_propose_launch(self.h, ctypes.byref(params)))
            self._queued.append(params)
            return
        # (the layout checks of a tensor are remembered per tensor object: between two searches the GPU waits for
        #  exactly this host code)
        ck = getattr(self, "_map_cache", None)
        if ck is None:
            ck = self._map_cache = {}
        ent = ck.get(id(fmap))
        ptr = fmap.data_ptr()
        if ent is None or ent[0] != ptr or ent[5]() is not fmap:
            import weakref
            t, cl = self._torch_map(fmap)
            C, H, W = (int(x) for x in t.shape)
            if len(ck) > 64:
                ck.clear()
            ent = ck[id(fmap)] = (ptr, C, H, W, 1 if cl else 0, weakref.ref(fmap), t.device)
        _, C, H, W, cl, _, dev = ent
        if producer_event is not None:
            self.wait_event(producer_event, params)
        elif not producer_done:
            import torch
            torch.cuda.current_stream(dev).synchronize()
        self._chk(self.L.az_propose_launch_on(self.h, ctypes.byref(params), ctypes.c_void_p(ptr), C, H, W, cl))
        self._queued.append(params)
        # (up to six searches may be queued -- three per lane: their maps stay referenced until they are long fetched)
        import collections
        self.__dict__.setdefault("_feat_keep", collections.deque(maxlen=10)).append(fmap)
        self._feat_keepalive = fmap
        self.feat_shape = (C, H, W)

    # ---- a batch of images in lockstep (az_batch_launch) ------------------------------------------------------------
    def batch_launch(self, params, fmaps, producer_done=False, producer_event=None):
        """The images of consecutive iterations of the dataset loop (lib/detect/test.py:508-513), all of ONE shape, searched
        together: every level's rois of all of them in one head pass (az_batch_launch).  fmaps: CUDA tensors [1,C,H,W] /
        [C,H,W] on this GPU, one per image (torch.channels_last ones are read where they lie, others are converted by torch);
        they must stay untouched until the batch's last batch_fetch.  producer_done / producer_event as propose_launch.
        Results: batch_fetch(i), i = 0 .. len(fmaps)-1 in order; each is what propose gives for that image alone.
        params may also be a LIST of AzParams, one per image: the images of the batch then have shapes of their own
        (az_batch_launch_shapes; they must walk the same number of levels and share num_proposals, eps, min_side, flags)."""
        import torch
        maps, ptrs, shape, hw = [], [], None, []
        converted = False
        # (the layout checks of a tensor are remembered per tensor object, as propose_launch does: a batch of 32 maps is
        #  otherwise ~0.3 ms of Python)
        ck = self.__dict__.setdefault("_bmap_cache", {})
        for f in fmaps:
            ent = ck.get(id(f))
            ptr = f.data_ptr()
            if ent is None or ent[0] != ptr or ent[2]() is not f:
                import weakref
                t = f if f.dim() == 4 else f[None]
                assert t.is_cuda and t.dtype == torch.float32 and t.device.index == self.device and t.shape[0] == 1
                if not t.is_contiguous(memory_format=torch.channels_last):
                    t = t.contiguous(memory_format=torch.channels_last)
                    converted = True
                    ent = None                       # (a converted copy is made again every time: the source may have changed)
                else:
                    # (only the checks' outcome is remembered, never the tensor or a view of it: an entry must not keep a
                    #  conv map alive -- a dataset loop hands over a fresh tensor per image; entries whose tensor has died
                    #  are dropped as soon as a few have gathered)
                    if len(ck) >= 64:
                        for k in [k for k, e in ck.items() if e[2]() is None]:
                            del ck[k]
                        if len(ck) >= 256:
                            ck.clear()
                    ent = ck[id(f)] = (ptr, tuple(int(x) for x in t.shape[1:]), weakref.ref(f))
                chw = tuple(int(x) for x in t.shape[1:])
            else:
                t, chw = (f if f.dim() == 4 else f[None]), ent[1]
            per_image = isinstance(params, (list, tuple))
            assert per_image or shape in (None, chw), "the maps of a batch have one shape (or pass one AzParams per image)"
            assert shape is None or shape[0] == chw[0]
            shape = chw
            maps.append(t)
            ptrs.append(t.data_ptr())
            hw.append(chw[1:])
        C, H, W = shape
        dev = maps[0].device
        if producer_event is not None and not converted:
            self._ext(self.L.az_batch_next_stream(self.h)).wait_event(producer_event)
        elif converted or not producer_done:
            torch.cuda.current_stream(dev).synchronize()
        arr = (ctypes.c_void_p * len(ptrs))(*ptrs)
        self._batch_stream = self._ext(self.L.az_batch_next_stream(self.h))       # (the stream this batch runs on)
        if isinstance(params, (list, tuple)):
            assert len(params) == len(ptrs)
            pa = (AzParams * len(ptrs))(*params)
            Hs = (ctypes.c_int * len(ptrs))(*[h for h, _ in hw])
            Ws = (ctypes.c_int * len(ptrs))(*[w for _, w in hw])
            self._chk(self.L.az_batch_launch_shapes(self.h, len(ptrs), pa, arr, C, Hs, Ws))
            params = params[0]
        else:
            self._chk(self.L.az_batch_launch(self.h, len(ptrs), ctypes.byref(params), arr, C, H, W))
        import collections
        q = self.__dict__.setdefault("_batches", collections.deque())
        q.append((params, maps))
        self.feat_shape = (C, H, W)

    def batch_stage_results(self, dst_ptr, pitch_bytes, cap_bytes):
        """Right behind batch_launch: the batch's result records to dst_ptr + i * pitch_bytes (a raw device pointer, e.g. rows
        of the RCCL send buffer), device to device; image i's is complete when batch_fetch(i) returns."""
        self._chk(self.L.az_batch_stage_results_dev(self.h, ctypes.c_void_p(int(dst_ptr)), int(pitch_bytes), int(cap_bytes)))

    def batch_fetch_all(self, want_scores=False, want_stats=False):
        """Every image of the oldest unfetched batch in one call (az_batch_fetch_all): a list of what batch_fetch(i) returns."""
        if not getattr(self, "_batches", None):
            raise AzError(-4, "batch_fetch_all without batch_launch")
        params, maps = self._batches[0]
        n, cap = len(maps), params.num_proposals
        boxes = np.empty((n, cap, 4), dtype=np.float64)
        scores = np.empty((n, cap), dtype=np.float32)
        cnt = (ctypes.c_int * n)()
        st = (AzStats * n)()
        try:
            self._chk(self.L.az_batch_fetch_all(self.h, _p(boxes, ctypes.c_double), _p(scores, ctypes.c_float), cap, cnt, st))
        finally:
            self._batches.popleft()
        out = []
        for i in range(n):
            r = [boxes[i, :cnt[i]].copy()]
            if want_scores:
                r.append(scores[i, :cnt[i]].copy())
            if want_stats:
                r.append(st[i])
            out.append(r[0] if len(r) == 1 else tuple(r))
        return out

    def batch_drain(self):
        """Collect and drop every batch still in flight (after an error in the caller's loop: the context is usable again)."""
        while getattr(self, "_batches", None):
            try:
                self.batch_fetch_all()
            except AzError:
                pass

    def batch_record_event(self):
        """A torch.cuda.Event recorded now on the stream of the batch launched last: behind that batch."""
        return self._batch_stream.record_event()

    def batch_fetch(self, i, want_scores=False, want_stats=False):
        if not getattr(self, "_batches", None):
            raise AzError(-4, "batch_fetch without batch_launch")
        params, maps = self._batches[0]
        cap = params.num_proposals
        boxes = np.empty((cap, 4), dtype=np.float64)
        scores = np.empty((cap,), dtype=np.float32)
        n = ctypes.c_int(0)
        st = AzStats()
        try:
            self._chk(self.L.az_batch_fetch(self.h, int(i), _p(boxes, ctypes.c_double), _p(scores, ctypes.c_float),
                                            cap, ctypes.byref(n), ctypes.byref(st)))
        finally:
            if i == len(maps) - 1:
                self._batches.popleft()
        out = [boxes[:n.value].copy()]
        if want_scores:
            out.append(scores[:n.value].copy())
        if want_stats:
            out.append(st)
        return out[0] if len(out) == 1 else tuple(out)

    def propose_fetch(self, want_scores=False, want_stats=False):
        params = self._queued.pop(0) if getattr(self, "_queued", None) else self._last_params
        cap = params.num_proposals if params.fixed_num else self.max_candidates
        boxes = np.empty((cap, 4), dtype=np.float64)
        scores = np.empty((cap,), dtype=np.float32)
        n = ctypes.c_int(0)
        st = AzStats()
        self._chk(self.L.az_propose_fetch(self.h, _p(boxes, ctypes.c_double), _p(scores, ctypes.c_float),
                                          cap, ctypes.byref(n), ctypes.byref(st)))
        out = [boxes[:n.value].copy()]
        if want_scores:
            out.append(scores[:n.value].copy())
        if want_stats:
            out.append(st)
        return out[0] if len(out) == 1 else tuple(out)

    @staticmethod
    def result_record_layout(num_proposals):
        """(bytes, n_offset, boxes_offset, scores_offset) of the device-resident result record of a
        fixed-count search (az_result_record_layout)."""
        L = load_library()
        v = [ctypes.c_size_t(0) for _ in range(4)]
        rc = L.az_result_record_layout(int(num_proposals), *[ctypes.byref(x) for x in v])
        if rc != AZ_OK:
            raise AzError(rc, "az_result_record_layout(%r)" % (num_proposals,))
        return tuple(int(x.value) for x in v)

    # ---- the exchange step, natively (one ncclAllGather on the ctx stream) ---------------------------------------------
    @staticmethod
    def rccl_unique_id():
        """128 bytes (made on rank 0, handed to the other ranks by the launcher) for rccl_init."""
        L = load_library()
        buf = ctypes.create_string_buffer(128)
        rc = L.az_rccl_unique_id(buf, 128)
        if rc != AZ_OK:
            raise AzError(rc, "az_rccl_unique_id: no usable librccl.so in this process")
        return bytes(buf.raw)

    def rccl_init(self, uid, nranks, rank):
        assert len(uid) == 128
        self._chk(self.L.az_rccl_init(self.h, ctypes.create_string_buffer(uid, 128), 128, int(nranks), int(rank)))

    def gather_records(self, send_ptr, recv_ptr, bytes_per_rank):
        """ncclAllGather of this rank's staged records (az_gather_records): enqueued on the ctx stream."""
        self._chk(self.L.az_gather_records(self.h, ctypes.c_void_p(int(send_ptr)), ctypes.c_void_p(int(recv_ptr)),
                                           int(bytes_per_rank)))

    def rccl_destroy(self):
        self._chk(self.L.az_rccl_destroy(self.h))

    def comm_stream(self):
        """torch view of the stream az_gather_records runs on (exists after rccl_init)."""
        return self._ext(self.L.az_comm_stream(self.h))

    def stage_result(self, dst_ptr, cap_bytes):
        """Between propose_launch and propose_fetch: enqueue a device-to-device copy of the result
        record to `dst_ptr` (a raw device pointer, e.g. a slot of the RCCL send buffer)."""
        self._chk(self.L.az_propose_stage_result_dev(self.h, ctypes.c_void_p(int(dst_ptr)), int(cap_bytes)))

    def last_candidates(self):
        cap = self.max_candidates
        boxes = np.empty((cap, 4), dtype=np.float64)
        scores = np.empty((cap,), dtype=np.float32)
        n = ctypes.c_int(0)
        self._chk(self.L.az_last_candidates(self.h, _p(boxes, ctypes.c_double), _p(scores, ctypes.c_float),
                                            cap, ctypes.byref(n)))
        return boxes[:n.value].copy(), scores[:n.value].copy()

    # ---- unit entry points ----------------------------------------------------------
    def divide_region(self, regions, min_side=10.0):
        regions = _f64(regions).reshape(-1, 4)
        cap = max(16, 12 * regions.shape[0] + 64)
        out = np.empty((cap, 4), dtype=np.float64)
        n = ctypes.c_int(0)
        rc = self.L.az_divide_region(self.h, _p(regions, ctypes.c_double), regions.shape[0], float(min_side),
                                     _p(out, ctypes.c_double), cap, ctypes.byref(n))
        if rc == AZ_ERR_CAPACITY and n.value > cap:
            cap = n.value
            out = np.empty((cap, 4), dtype=np.float64)
            rc = self.L.az_divide_region(self.h, _p(regions, ctypes.c_double), regions.shape[0],
                                         float(min_side), _p(out, ctypes.c_double), cap, ctypes.byref(n))
        self._chk(rc)
        return out[:n.value].copy()

    def sift_dup(self, regions, min_side=10.0):
        regions = _f64(regions).reshape(-1, 4)
        cap = max(1, regions.shape[0])
        out = np.empty((cap, 4), dtype=np.float64)
        n = ctypes.c_int(0)
        self._chk(self.L.az_sift_dup(self.h, _p(regions, ctypes.c_double), regions.shape[0], float(min_side),
                                     _p(out, ctypes.c_double), cap, ctypes.byref(n)))
        return out[:n.value].copy()

    def roi_dedup(self, boxes, scale, dedup=1. / 16., batch_size=10000):
        boxes = _f64(boxes).reshape(-1, 4)
        P = boxes.shape[0]
        rois = np.empty((max(P, 1), 5), dtype=np.float32)
        index = np.empty((max(P, 1),), dtype=np.int32)
        inv = np.empty((max(P, 1),), dtype=np.int32)
        n = ctypes.c_int(0)
        self._chk(self.L.az_roi_dedup(self.h, _p(boxes, ctypes.c_double), P, float(scale), float(dedup),
                                      int(batch_size), _p(rois, ctypes.c_float), _p(index, ctypes.c_int32),
                                      _p(inv, ctypes.c_int32), ctypes.byref(n)))
        return rois[:P].copy(), index[:n.value].copy(), inv[:P].copy()

    def roi_pool(self, rois):
        rois = _f32(rois).reshape(-1, 5)
        R = rois.shape[0]
        out = np.empty((max(R, 1), self.dims["K6"]), dtype=np.float32)
        self._chk(self.L.az_roi_pool(self.h, _p(rois, ctypes.c_float), R, _p(out, ctypes.c_float)))
        return out[:R]

    def head_forward(self, rois):
        rois = _f32(rois).reshape(-1, 5)
        R = rois.shape[0]
        z = np.empty((max(R, 1), 1), dtype=np.float32)
        p = np.empty((max(R, 1), AZ_NUM_SUBREG), dtype=np.float32)
        d = np.empty((max(R, 1), 4 * AZ_NUM_SUBREG), dtype=np.float32)
        self._chk(self.L.az_head_forward(self.h, _p(rois, ctypes.c_float), R, _p(z, ctypes.c_float),
                                         _p(p, ctypes.c_float), _p(d, ctypes.c_float)))
        return z[:R], p[:R], d[:R]

    def decode_filter(self, anchors, deltas, scores, im_h, im_w, eps=1e-14, min_side=10.0):
        anchors = _f64(anchors).reshape(-1, 4)
        R = anchors.shape[0]
        deltas = _f32(deltas).reshape(R, 4 * AZ_NUM_SUBREG)
        scores = _f32(scores).reshape(R, AZ_NUM_SUBREG)
        cap = max(1, R * AZ_NUM_SUBREG)
        ob = np.empty((cap, 4), dtype=np.float64)
        os_ = np.empty((cap,), dtype=np.float32)
        n = ctypes.c_int(0)
        self._chk(self.L.az_decode_filter(self.h, _p(anchors, ctypes.c_double), _p(deltas, ctypes.c_float),
                                          _p(scores, ctypes.c_float), R, int(im_h), int(im_w), float(eps),
                                          float(min_side), _p(ob, ctypes.c_double), _p(os_, ctypes.c_float),
                                          cap, ctypes.byref(n)))
        return ob[:n.value].copy(), os_[:n.value].copy()

    def topk(self, scores, k):
        scores = _f32(scores).ravel()
        idx = np.empty((max(1, min(k, scores.shape[0])),), dtype=np.int32)
        n = ctypes.c_int(0)
        self._chk(self.L.az_topk(self.h, _p(scores, ctypes.c_float), scores.shape[0], int(k),
                                 _p(idx, ctypes.c_int32), ctypes.byref(n)))
        return idx[:n.value].copy()

    def nms(self, dets, thresh):
        dets = _f32(dets).reshape(-1, 5)
        N = dets.shape[0]
        keep = np.empty((max(N, 1),), dtype=np.int64)
        n = ctypes.c_int(0)
        self._chk(self.L.az_nms(self.h, _p(dets, ctypes.c_float), N, float(thresh), _p(keep, ctypes.c_int64),
                                ctypes.byref(n)))
        return keep[:n.value].copy()

    # ---- Fast R-CNN head on the shared map -----------------------------------------------
    def nms_batched(self, dets_list, thresh):
        """NMS of many independent box sets ([n_g,5] float32 each) in one call -> list of keep index
        arrays (what apply_nms, lib/detect/test.py:467-484, needs per class per image)."""
        n = len(dets_list)
        off = np.zeros(n + 1, dtype=np.int32)
        for g, d in enumerate(dets_list):
            off[g + 1] = off[g] + d.shape[0]
        allb = _f32(np.vstack([np.zeros((0, 5), np.float32)] + [np.asarray(d, dtype=np.float32).reshape(-1, 5)
                                                                   for d in dets_list]))
        keep = np.zeros(max(int(off[n]), 1), dtype=np.int64)
        nk = np.zeros(max(n, 1), dtype=np.int32)
        self._chk(self.L.az_nms_batched(self.h, _p(allb, ctypes.c_float), _p(off, ctypes.c_int32), n, float(thresh),
                                        _p(keep, ctypes.c_int64), _p(nk, ctypes.c_int32)))
        return [keep[off[g]:off[g] + nk[g]].copy() for g in range(n)]

    def load_det_head(self, head):
        """head: dict of Caffe-layout fp32 arrays W6,b6 (fc6), W7,b7 (fc7), Wc,bc (cls_score),
        Wb,bb (bbox_pred)."""
        W6 = _f32(head["W6"])
        n6, K6 = W6.shape
        assert K6 % 49 == 0
        C = K6 // 49
        n7 = head["W7"].shape[0]
        ncls = head["Wc"].shape[0]
        assert head["W7"].shape == (n7, n6) and head["Wc"].shape == (ncls, n7) and head["Wb"].shape == (4 * ncls, n7)
        arrs = [W6] + [_f32(head[k]) for k in ("b6", "W7", "b7", "Wc", "bc", "Wb", "bb")]
        self._chk(self.L.az_load_det_head(self.h, C, n6, n7, ncls, *[_p(a, ctypes.c_float) for a in arrs]))
        self.det_dims = dict(C=C, n6=n6, n7=n7, ncls=ncls)

    def det_forward(self, rois):
        rois = _f32(rois).reshape(-1, 5)
        R = rois.shape[0]
        nc = self.det_dims["ncls"]
        p = np.empty((max(R, 1), nc), dtype=np.float32)
        b = np.empty((max(R, 1), 4 * nc), dtype=np.float32)
        self._chk(self.L.az_det_forward(self.h, _p(rois, ctypes.c_float), R, _p(p, ctypes.c_float),
                                        _p(b, ctypes.c_float)))
        return p[:R], b[:R]

    def detect(self, boxes, scale, im_h, im_w, dedup=1. / 16., batch_size=10000, eps=1e-14):
        boxes = _f64(boxes).reshape(-1, 4)
        P = boxes.shape[0]
        nc = self.det_dims["ncls"]
        s = np.empty((max(P, 1), nc), dtype=np.float32)
        b = np.empty((max(P, 1), 4 * nc), dtype=np.float64)
        self._chk(self.L.az_detect(self.h, _p(boxes, ctypes.c_double), P, float(scale), float(dedup),
                                   int(batch_size), int(im_h), int(im_w), float(eps), _p(s, ctypes.c_float),
                                   _p(b, ctypes.c_double)))
        return s[:P], b[:P]

    # ---- tuner ------------------------------------------------------------------------
    def last_anchors(self):
        """(regions [n,4] f64, zoom [n] f32) of the last tuner-variant search (Bhis, tune.py:303)."""
        cap = 2 * self.max_regions
        regions = np.empty((cap, 4), dtype=np.float64)
        zoom = np.empty((cap,), dtype=np.float32)
        n = ctypes.c_int(0)
        self._chk(self.L.az_last_anchors(self.h, _p(regions, ctypes.c_double), _p(zoom, ctypes.c_float), cap,
                                         ctypes.byref(n)))
        return regions[:n.value].copy(), zoom[:n.value].copy()

    def tune_begin(self, capacity):
        self._chk(self.L.az_tune_begin(self.h, int(capacity)))

    def tune_end(self):
        self._chk(self.L.az_tune_end(self.h))

    def tune_kth_largest(self, k):
        """(k-th largest pooled zoom score as float, number of pooled scores); -inf when <= k."""
        v = ctypes.c_float(0)
        n = ctypes.c_longlong(0)
        self._chk(self.L.az_tune_kth_largest(self.h, int(k), ctypes.byref(v), ctypes.byref(n)))
        return float(v.value), int(n.value)

    def tune_top(self, k):
        cap = int(k) + 65536
        while True:
            out = np.empty((cap,), dtype=np.float32)
            n = ctypes.c_longlong(0)
            rc = self.L.az_tune_top(self.h, int(k), _p(out, ctypes.c_float), cap, ctypes.byref(n))
            if rc == AZ_ERR_CAPACITY and n.value > cap:
                cap = int(n.value)
                continue
            self._chk(rc)
            return out[:n.value].copy()

    def tune_push(self, scores):
        a = _f32(scores).ravel()
        self._chk(self.L.az_tune_push(self.h, _p(a, ctypes.c_float), a.size))

    # ---- recall evaluation -------------------------------------------------------------
    def bbox_overlaps(self, boxes, query_boxes):
        b = _f64(boxes).reshape(-1, 4)
        q = _f64(query_boxes).reshape(-1, 4)
        out = np.zeros((b.shape[0], q.shape[0]), dtype=np.float64)
        self._chk(self.L.az_bbox_overlaps(self.h, _p(b, ctypes.c_double), b.shape[0], _p(q, ctypes.c_double),
                                          q.shape[0], _p(out, ctypes.c_double)))
        return out

    def recall_match(self, boxes_list, gt_list):
        """Per-image greedy matching of imdb.evaluate_recall for lists of [n_i,4] candidate and
        [k_i,4] ground-truth boxes -> concatenated gt overlaps (image order, then pick order)."""
        assert len(boxes_list) == len(gt_list)
        n = len(boxes_list)
        boff = np.zeros(n + 1, dtype=np.int32)
        goff = np.zeros(n + 1, dtype=np.int32)
        for i in range(n):
            boff[i + 1] = boff[i] + boxes_list[i].shape[0]
            goff[i + 1] = goff[i] + gt_list[i].shape[0]
        b = _f64(np.vstack([np.zeros((0, 4))] + [x.reshape(-1, 4) for x in boxes_list]))
        g = _f64(np.vstack([np.zeros((0, 4))] + [x.reshape(-1, 4) for x in gt_list]))
        out = np.zeros((int(goff[n]),), dtype=np.float64)
        self._chk(self.L.az_recall_match(self.h, n, _p(b, ctypes.c_double), _p(boff, ctypes.c_int32),
                                         _p(g, ctypes.c_double), _p(goff, ctypes.c_int32), _p(out, ctypes.c_double)))
        return out

    # ---- image front-end ---------------------------------------------------------------
    def image_blob_size(self, h, w, scale):
        oh, ow = ctypes.c_int(0), ctypes.c_int(0)
        rc = self.L.az_image_blob_size(int(h), int(w), float(scale), ctypes.byref(oh), ctypes.byref(ow))
        if rc != AZ_OK:
            raise AzError(rc, "az_image_blob_size: bad arguments")
        return oh.value, ow.value

    def image_blob(self, im, means, scale, out=None, stream=None):
        """uint8 BGR HWC image -> [1,3,oh,ow] f32 blob (mean-subtracted, cv2-style bilinear).
        out: None -> NumPy array; a CUDA torch tensor of the right shape -> filled in place.
        stream (with a tensor `out`): a raw hipStream_t handle, e.g. torch.cuda.current_stream().cuda_stream -- upload and
        kernel are only ENQUEUED there (az_image_blob_dev_on), ordered with whatever that stream runs next."""
        im = np.ascontiguousarray(im, dtype=np.uint8)
        assert im.ndim == 3 and im.shape[2] == 3
        h, w = im.shape[:2]
        oh, ow = self.image_blob_size(h, w, scale)
        m = _f32(np.asarray(means).ravel())
        assert m.size == 3
        if out is None:
            blob = np.empty((1, 3, oh, ow), dtype=np.float32)
            self._chk(self.L.az_image_blob_host(self.h, _p(im, ctypes.c_uint8), h, w, _p(m, ctypes.c_float),
                                                float(scale), _p(blob, ctypes.c_float), oh, ow))
            return blob
        assert tuple(out.shape[-3:]) == (3, oh, ow) and out.is_contiguous() and out.is_cuda
        if stream is not None:
            # (torch's default stream has the handle 0, which az_image_blob_dev_on reads as "the ctx stream": the default
            #  stream is named explicitly -- hipStreamLegacy, (hipStream_t)1 -- or the upload and the front-end kernel would
            #  run on another stream than the backbone that reads the blob)
            sh = int(stream) or 1
            self._chk(self.L.az_image_blob_dev_on(self.h, _p(im, ctypes.c_uint8), h, w, _p(m, ctypes.c_float), float(scale),
                                                  ctypes.c_void_p(out.data_ptr()), oh, ow, ctypes.c_void_p(sh)))
            return out
        self._chk(self.L.az_image_blob_dev(self.h, _p(im, ctypes.c_uint8), h, w, _p(m, ctypes.c_float),
                                           float(scale), ctypes.c_void_p(out.data_ptr()), oh, ow))
        return out

    # ---- measurement -----------------------------------------------------------------
    def set_profiling(self, mode):
        """mode bits: 1 = fc GEMM launches only, 2 = every launch group, 4 = accumulate across
        calls until read; 0 = off."""
        self._chk(self.L.az_set_profiling(self.h, int(mode)))

    def set_graphs(self, on):
        """Replay the search's launch sequence as a hipGraph (same results, less host time)."""
        self._chk(self.L.az_set_graphs(self.h, 1 if on else 0))

    # the round-3 profile's figures for the full head on one MI355X box: what tests pin the form choice to
    REFERENCE_PASS_COSTS = ((40, 142.0), (704, 1096.0))

    def set_pass_costs(self, table=None):
        """table: ((rows, us), ...) ascending, 2..6 points -- the cost of one head pass the context chooses the search
        form by (az_set_pass_costs); None / () = measure on the device at the next launch (the default)."""
        t = list(table or ())
        rows = np.ascontiguousarray([r for r, _ in t], dtype=np.int32)
        us = np.ascontiguousarray([u for _, u in t], dtype=np.float64)
        self._chk(self.L.az_set_pass_costs(self.h, len(t), _p(rows, ctypes.c_int32) if t else None,
                                           _p(us, ctypes.c_double) if t else None))

    def pass_costs(self):
        """((rows, us), ...) in use; () before the first launch of a context that measures its own."""
        rows = np.zeros(8, dtype=np.int32)
        us = np.zeros(8, dtype=np.float64)
        n = ctypes.c_int(0)
        self._chk(self.L.az_get_pass_costs(self.h, _p(rows, ctypes.c_int32), _p(us, ctypes.c_double), 8, ctypes.byref(n)))
        return tuple((int(rows[i]), float(us[i])) for i in range(n.value))

    def measure_box(self):
        """(fp32 MFMA TFLOP/s this box sustains in a register-only loop, TB/s of a 1 GiB float4 copy) -- az_measure_box."""
        a, b = ctypes.c_double(0), ctypes.c_double(0)
        self._chk(self.L.az_measure_box(self.h, ctypes.byref(a), ctypes.byref(b)))
        return float(a.value), float(b.value)

    def last_kernel_times(self, cap=65536):
        names = ctypes.create_string_buffer(32 * cap)
        ms = np.zeros(cap, dtype=np.float32)
        lv = np.zeros(cap, dtype=np.int32)
        n = ctypes.c_int(0)
        self._chk(self.L.az_last_kernel_times(self.h, names, _p(ms, ctypes.c_float), _p(lv, ctypes.c_int32),
                                              cap, ctypes.byref(n)))
        out = []
        for i in range(min(n.value, cap)):
            nm = names.raw[32 * i:32 * i + 32].split(b"\0", 1)[0].decode()
            out.append((nm, int(lv[i]), float(ms[i])))
        return out

    def stream_handle(self):
        return self.L.az_stream(self.h)


_default_ctx = None


def bias_relu_(y, bias):
    """In place on a CUDA fp32 tensor y [1,C,H,W] (contiguous or channels_last): y = max(y + bias[c], 0) in one launch on
    torch's current stream (az_bias_relu).  Returns y."""
    import torch
    L = load_library()
    C, H, W = int(y.shape[1]), int(y.shape[2]), int(y.shape[3])
    cl = y.is_contiguous(memory_format=torch.channels_last) and not (y.is_contiguous() and C > 1 and H * W > 1)
    if not cl and not y.is_contiguous():
        raise ValueError("bias_relu_: the tensor must be contiguous or channels_last")
    rc = L.az_bias_relu(ctypes.c_void_p(torch.cuda.current_stream(y.device).cuda_stream), ctypes.c_void_p(y.data_ptr()),
                        ctypes.c_void_p(bias.data_ptr()), C, H * W, 1 if cl else 0)
    if rc != AZ_OK:
        raise AzError(rc, "az_bias_relu")
    return y


def bias_relu_pool(y, bias):
    """y [1,C,H,W] CUDA fp32 (contiguous or channels_last) -> max_pool2d(relu(y + bias), 2, 2, ceil_mode=True) as a new tensor
    of the same memory format, one launch on torch's current stream (az_bias_relu_pool)."""
    import torch
    L = load_library()
    C, H, W = int(y.shape[1]), int(y.shape[2]), int(y.shape[3])
    cl = y.is_contiguous(memory_format=torch.channels_last) and not (y.is_contiguous() and C > 1 and H * W > 1)
    if not cl and not y.is_contiguous():
        raise ValueError("bias_relu_pool: the tensor must be contiguous or channels_last")
    out = torch.empty((1, C, (H + 1) // 2, (W + 1) // 2), dtype=torch.float32, device=y.device,
                      memory_format=torch.channels_last if cl else torch.contiguous_format)
    rc = L.az_bias_relu_pool(ctypes.c_void_p(torch.cuda.current_stream(y.device).cuda_stream), ctypes.c_void_p(y.data_ptr()),
                             ctypes.c_void_p(bias.data_ptr()), ctypes.c_void_p(out.data_ptr()), C, H, W, 1 if cl else 0)
    if rc != AZ_OK:
        raise AzError(rc, "az_bias_relu_pool")
    return out


def _rank_device():
    """This process's GPU: torch's current device when torch has one, else LOCAL_RANK, else 0 --
    one process per GPU, so the drop-in helpers must not all land on GPU 0."""
    try:
        import torch
        if torch.cuda.is_available():
            return int(torch.cuda.current_device())
    except Exception:
        pass
    return int(os.environ.get("LOCAL_RANK", "0"))


def set_default_context(ctx):
    """Make `ctx` (the context of this rank's HipAZNet) the one the drop-in modules use."""
    global _default_ctx
    _default_ctx = ctx
    return ctx


def default_context(device=None):
    """Process-wide context used by the drop-in modules (utils.cython_div / cython_nms / cython_bbox,
    apply_nms): the rank's own net's context when one exists (HipAZNet registers itself), otherwise a
    context created on this rank's GPU."""
    global _default_ctx
    if _default_ctx is None or getattr(_default_ctx, "h", None) is None:
        _default_ctx = AzContext(_rank_device() if device is None else device)
    return _default_ctx
