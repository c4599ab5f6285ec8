/*
 * aznet_hip.h -- C ABI of libaznet_hip.so: the MI355X (gfx950) implementation of
 * AZ-Net's adjacency-and-zoom region-proposal search.
 *
 * The reference (luyongxi/az-net) has no C API: its native surface is three Cython
 * modules plus pycaffe.  Each entry point below names the reference interface it
 * replaces (paths relative to the upstream tree).  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns AZ_OK (0) or a negative az_status; az_last_error(ctx)
 *     gives a message for the last failure on that context.  No exceptions and no
 *     C++ types cross the boundary.
 *   - all pointers are HOST pointers unless the parameter says "dev"; outputs are
 *     caller-allocated with an explicit capacity and an out-count.
 *   - one az_ctx per GPU; a ctx is not thread-safe, distinct ctxs are independent.
 *   - calls are synchronous for the caller; inside they are stream-ordered HIP work
 *     with a single host synchronisation at the end (az_propose: none inside the
 *     level loop).
 *   - boxes are (x1, y1, x2, y2) float64 in ORIGINAL image pixels, as in
 *     lib/detect/test.py:346-414; rois are (batch, x1, y1, x2, y2) float32 in scaled
 *     image pixels, as in lib/detect/test.py:61-71.
 */
#ifndef AZNET_HIP_H
#define AZNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct az_ctx az_ctx;

typedef enum {
    AZ_OK = 0,
    AZ_ERR_INVALID = -1,      /* bad argument (NULL, negative size, shape mismatch) */
    AZ_ERR_HIP = -2,          /* a HIP runtime call failed; see az_last_error       */
    AZ_ERR_CAPACITY = -3,     /* a level / candidate list outgrew the ctx limits or `cap` */
    AZ_ERR_STATE = -4,        /* head or feature map not loaded                      */
    AZ_ERR_NO_DEVICE = -5     /* no usable gfx950 device: there is NO CPU fallback   */
} az_status;

#define AZ_MAX_LEVELS 16
#define AZ_NUM_SUBREG 11      /* len(cfg.SEAR.SUBREGION), lib/detect/config.py:149-155 */

/* Search parameters = the cfg keys lib/detect/test.py reads on this path. */
typedef struct {
    int32_t im_h, im_w;       /* original image size (im.shape[0:2])                  */
    double  scale;            /* im_scale of the single test scale (test.py:45-50)     */
    double  Tz;               /* cfg.SEAR.Tz  (config.py:272-280); compared in double  */
    double  Tc;               /* cfg.SEAR.Tc  (config.py:171), used when !fixed_num    */
    double  dedup;            /* cfg.DEDUP_BOXES = 1/16 (config.py:206)                */
    double  eps;              /* cfg.EPS = 1e-14 (config.py:216)                       */
    double  min_side;         /* cfg.SEAR.MIN_SIDE = 10 (config.py:186)                */
    int32_t batch_size;       /* cfg.SEAR.BATCH_SIZE (config.py:189): dedup chunk size */
    int32_t num_proposals;    /* cfg.SEAR.NUM_PROPOSALS (config.py:133, 279)           */
    int32_t fixed_num;        /* cfg.SEAR.FIXED_PROPOSAL_NUM (config.py:172)           */
    int32_t reserved;         /* flags; bit 0: evaluate levels 1-3 one by one (no speculation);
                                 bit 1: keep their geometry as separate launches (same bits)   */
} az_params;

/* What the reference prints per image (test.py:408-409) plus per-level sizes. */
typedef struct {
    int32_t n_proposals;
    int32_t num_eval;                     /* sum of B.shape[0] over levels (test.py:378) */
    int32_t depth;                        /* last k of the level loop                    */
    int32_t n_levels;                     /* K - 1                                       */
    int32_t n_candidates;                 /* len(aScores) before selection               */
    int32_t level_regions[AZ_MAX_LEVELS]; /* B.shape[0] per level                        */
    int32_t level_unique[AZ_MAX_LEVELS];  /* rois actually forwarded (after 1/16 dedup)  */
    int32_t level_zoomed[AZ_MAX_LEVELS];  /* len(indZ)                                   */
} az_stats;

/* ---- lifecycle ----------------------------------------------------------------- */
const char *az_version(void);
/* Replaces caffe.set_mode_gpu(); caffe.set_device(id) (tools/prop_az.py:88-89). */
int az_create(int device, az_ctx **out);
int az_destroy(az_ctx *ctx);
const char *az_last_error(const az_ctx *ctx);
/* Optional, before az_load_head: per-level region capacity (default 16384) and total
 * candidate capacity (default 16384*11).  Buffers are sized once, for 288 GB of HBM. */
int az_set_limits(az_ctx *ctx, int max_regions, int max_candidates);

/* Optional, before az_load_head.  How int6 (95 % of the head's FLOPs) is evaluated for launches of
 * more than 64 rois:
 *   0 (default)  fp32 MFMA (v_mfma_f32_32x32x2_f32), bitwise an fmaf chain;
 *   3            fp32 operands split into three bf16 round-off terms, six bf16 MFMAs per product
 *                with fp32 accumulation: products good to 2^-24 (fp32-grade), 6/16 of the cost;
 *   2            two terms, three MFMAs: products good to ~2^-16 (outputs still within 1e-4), 3/16.
 * Launches of <= 64 rois are weight-streaming bound and always use the fp32 kernel. */
int az_set_gemm_mode(az_ctx *ctx, int parts);

/* Replaces caffe.Net(test_fc.prototxt, caffemodel) (tools/prop_az.py:95-96): the AZ head
 * models/Pascal/VGG16/az-net/test_fc.prototxt:14-232.  Weights are Caffe InnerProduct
 * blobs, row-major [out, in]; they are copied (and re-tiled) into HBM.
 *   W6 [n6, C*49] b6 [n6] | W71 [n71, n6] b71 | W72 [n72, n6] b72
 *   Was [11, n71] bas | Wab [44, n71] bab | Wz [1, n72] bz
 * C, n6 must be multiples of 4. */
int az_load_head(az_ctx *ctx, int C, int n6, int n71, int n72,
                 const float *W6, const float *b6, const float *W71, const float *b71,
                 const float *W72, const float *b72, const float *Was, const float *bas,
                 const float *Wab, const float *bab, const float *Wz, const float *bz);

/* Replaces feeding `conv5_3` to net['fc'] (lib/detect/test.py:229-236).  The map (NCHW f32,
 * batch 1) is transposed once into ctx-owned HBM in the channel-last layout RoIPool reads;
 * the call returns after that copy, so the source may be reused or freed afterwards.
 * _dev: device pointer, e.g. a torch tensor's data_ptr(); the producer's stream must have
 *       finished writing it before the call.
 * _host: host array. */
int az_set_feature_map_dev(az_ctx *ctx, const float *dev_ptr, int C, int H, int W);
int az_set_feature_map_host(az_ctx *ctx, const float *host_ptr, int C, int H, int W);

/* ---- the hot path --------------------------------------------------------------- */
/* Replaces im_propose (lib/detect/test.py:346-414) given the cached conv5_3: the whole
 * level loop (roi projection + 1/16 dedup, RoIPool, fc head, sigmoid, box decode, clip,
 * MIN_SIDE filter, zoom select, divide_region + _sift_dup, final top-K / Tc select) runs
 * on the GPU.  boxes_out [cap,4] f64, scores_out [cap] f32 (may be NULL). */
int az_propose(az_ctx *ctx, const az_params *p, double *boxes_out, float *scores_out,
               int cap, int *n_out, az_stats *stats);
/* Same search split in two so the caller can overlap other GPU work (the next image's
 * backbone): _launch enqueues everything and returns, _fetch waits and copies out. */
int az_propose_launch(az_ctx *ctx, const az_params *p);
int az_propose_fetch(az_ctx *ctx, double *boxes_out, float *scores_out, int cap, int *n_out,
                     az_stats *stats);
/* All candidates of the last az_propose, before selection (Y / aScores of test.py:380-381). */
int az_last_candidates(az_ctx *ctx, double *boxes_out, float *scores_out, int cap, int *n_out);

/* ---- unit entry points (the same kernels, one stage at a time) --------------------- */
/* utils.cython_div.divide_region(regions f64[P,4], min_height) (lib/utils/div.pyx:15-76). */
int az_divide_region(az_ctx *ctx, const double *regions, int P, double min_side,
                     double *out, int cap, int *n_out);
/* utils.cython_div._sift_dup (lib/utils/div.pyx:78-89). */
int az_sift_dup(az_ctx *ctx, const double *regions, int C, double min_side,
                double *out, int cap, int *n_out);
/* _get_rois_blob + the feature-space dedup of lib/detect/test.py:61-97,210-218 for one
 * level.  rois_out [P,5] f32 (all rois, before dedup), index_out [P] (first n_unique
 * valid), inv_index_out [P]. */
int az_roi_dedup(az_ctx *ctx, const double *boxes, int P, double scale, double dedup,
                 int batch_size, float *rois_out, int32_t *index_out, int32_t *inv_index_out,
                 int *n_unique);
/* Caffe ROIPooling 7x7 @ spatial_scale (test_fc.prototxt:14-25) over the current
 * feature map.  out [R, C*49] f32. */
int az_roi_pool(az_ctx *ctx, const float *rois, int R, float *out);
/* net['fc'].forward(rois=...) (lib/detect/test.py:235-242): zoom_prob [R,1],
 * adj_prob [R,11], adj_bbox [R,44], all f32.  Any output may be NULL. */
int az_head_forward(az_ctx *ctx, const float *rois, int R, float *zoom_prob, float *adj_prob,
                    float *adj_bbox);
/* _bbox_pred + _clip_boxes + _unwrap_adj_pred (lib/detect/test.py:106-151,171-187) for R
 * regions: anchors [R,4] f64, deltas [R,44] f32, scores [R,11] f32 -> kept boxes/scores in
 * r*11+s order. */
int az_decode_filter(az_ctx *ctx, const double *anchors, const float *deltas, const float *scores,
                     int R, int im_h, int im_w, double eps, double min_side,
                     double *boxes_out, float *scores_out, int cap, int *n_out);
/* Final selection of lib/detect/test.py:393-401: indices of the top-k scores, descending
 * (ties: lower index first). */
int az_topk(az_ctx *ctx, const float *scores, int n, int k, int32_t *idx_out, int *n_out);
/* utils.cython_nms.nms(dets f32[N,5], thresh) (lib/utils/nms.pyx:17-68): kept original
 * indices in descending-score order.  The reference's call site is apply_nms
 * (lib/detect/test.py:467-484); it is NOT on the proposal path. */
int az_nms(az_ctx *ctx, const float *dets, int n, double thresh, int64_t *keep, int *n_keep);

/* ---- Fast R-CNN head on the shared conv map (BASELINE config 3) ---------------------- */
/* Replaces caffe.Net(frcnn/test_fc.prototxt, caffemodel) (tools/test_shared.py): the detection
 * head models/Pascal/VGG16/frcnn/test_fc.prototxt:14-145 -- fc6 [n6, C*49], fc7 [n7, n6],
 * cls_score [ncls, n7] (+Softmax), bbox_pred [4*ncls, n7]; Caffe [out, in] layout. */
int az_load_det_head(az_ctx *ctx, int C, int n6, int n7, int ncls, const float *W6, const float *b6,
                     const float *W7, const float *b7, const float *Wc, const float *bc,
                     const float *Wb, const float *bb);
/* frcnn_net['fc'].forward(rois=..., conv5_3=...) (lib/detect/test.py:302-307): cls_prob [R,ncls],
 * bbox_pred [R,4*ncls], f32. */
int az_det_forward(az_ctx *ctx, const float *rois, int R, float *cls_prob, float *bbox_pred);
/* _frcnn_forward (lib/detect/test.py:259-318) for the proposals `boxes` [P,4] f64 of one image:
 * roi projection + 1/16 dedup per batch_size chunk, head, _bbox_pred + _clip_boxes of every
 * class, un-dedup.  scores_out [P,ncls] f32, boxes_out [P,4*ncls] f64. */
int az_detect(az_ctx *ctx, const double *boxes, int P, double scale, double dedup, int batch_size,
              int im_h, int im_w, double eps, float *scores_out, double *boxes_out);

/* ---- measurement ------------------------------------------------------------------ */
/* HIP-event timing (events on the ctx stream) of the launches made by az_propose /
 * az_head_forward.  mode bits: 1 = time only the fc GEMM launches, 2 = time every launch
 * group, 4 = keep accumulating across calls until read (otherwise each call starts afresh);
 * 0 = off.  names_out: `cap` slots of 32 chars. */
int az_set_profiling(az_ctx *ctx, int mode);
int az_last_kernel_times(az_ctx *ctx, char *names_out, float *ms_out, int32_t *level_out,
                         int cap, int *n_out);
/* The HIP stream the ctx launches on (a hipStream_t). */
void *az_stream(az_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* AZNET_HIP_H */
