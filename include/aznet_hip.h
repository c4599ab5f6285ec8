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
#define AZ_BATCH_MAX 32        /* images az_batch_launch searches in lockstep */

/* Search parameters = the cfg keys lib/detect/test.py reads on this path. */
typedef struct {
    int32_t im_h, im_w;       /* original image size (im.shape[0:2])                  */
    double  scale;            /* im_scale of the single test scale (test.py:45-50)     */
    double  Tz;               /* cfg.SEAR.Tz  (config.py:272-280); compared in double  */
    double  Tc;               /* cfg.SEAR.Tc  (config.py:171), used when !fixed_num    */
    double  dedup;            /* cfg.DEDUP_BOXES = 1/16 (config.py:206); <= 0: no dedup
                                 (test.py:211 `if cfg.DEDUP_BOXES > 0:`): every region forwarded */
    double  eps;              /* cfg.EPS = 1e-14 (config.py:216)                       */
    double  min_side;         /* cfg.SEAR.MIN_SIDE = 10 (config.py:186)                */
    int32_t batch_size;       /* cfg.SEAR.BATCH_SIZE (config.py:189): dedup chunk size */
    int32_t num_proposals;    /* cfg.SEAR.NUM_PROPOSALS (config.py:133, 279)           */
    int32_t fixed_num;        /* cfg.SEAR.FIXED_PROPOSAL_NUM (config.py:172)           */
    int32_t reserved;         /* flags; bit 0: evaluate levels 1-3 one by one (no speculation);
                                 bit 1: keep their geometry as separate launches (same bits);
                                 bit 2: the tuner's variant of the search (lib/detect/tune.py:
                                 256-316): K levels instead of K-1, Tz applied from the second
                                 level on, root not forced, anchor history kept (az_last_anchors);
                                 bit 3: final top-k by the single-workgroup radix select instead of
                                 the chip-wide counting kernels (same result; for tests);
                                 bit 4: keep the geometry of the levels after the speculative ones as
                                 separate launches instead of one kernel per level (same bits);
                                 bit 5: with Tz <= 0 still walk the tree level by level (by default such a
                                 search -- every finite zoom score passes `zoom >= Tz`, test.py:386, so the
                                 tree depends on the image shape only -- forwards the rois of ALL levels in
                                 ONE head pass; same bits);
                                 bit 6: no pair speculation; bit 7: pair speculation at every eligible level (by
                                 default the context decides from its previous search whether the head pass of a
                                 level also evaluates the rois of ALL children of its regions, a superset of the next
                                 level's, which then needs no pass of its own; same bits in all three);
                                 bit 8: no whole-tree speculation; bit 9: whole-tree speculation whenever the image
                                 shape allows (by default the context decides from the row counts of its previous
                                 search of the shape: a dense tree's ONE head pass evaluates a shape-static superset
                                 of its rows and every level finds its outputs by RoIPool window -- after a full tree
                                 the full tree's unique rois, where a search that needs a window the pass lacks is
                                 repeated level by level; otherwise the closure rows, which hold every region any
                                 pruning can produce; same bits in all three);
                                 bit 10: with bit 9, the closure rows instead of the full tree's;
                                 bit 12: no early end (by default, when the context's previous search of the image shape
                                 had no regions from some level on AND its last four searches all ended there or earlier,
                                 a search is enqueued only up to that level -- an empty level still costs its launches,
                                 ~35 us -- and is run again in full if this tree goes on; same bits)                 */
} az_params;

/* The bits of az_params.reserved by name.  A caller that just wants the search passes 0.  AZ_P_TUNE is the ONLY bit that
 * changes a result (it selects the tuner's variant of the search, lib/detect/tune.py:256-316); every other bit picks among
 * forms of the same search that give the same bits -- they exist for the parity tests (each form against the plain level
 * loop) and for A/B measurements, and the context's own choice (all bits 0) is what bench.py and the tools run. */
enum {
    AZ_P_NO_SPECULATION      = 1,     /* levels 1-3 one head pass each                                  */
    AZ_P_UNFUSED_FIRST_LEVELS = 2,    /* their geometry as separate launches                            */
    AZ_P_TUNE                = 4,     /* the tuner's variant (K levels, no forced root, anchor history) */
    AZ_P_RADIX_SELECT        = 8,     /* final top-k by the single-workgroup radix select               */
    AZ_P_UNFUSED_LEVELS      = 16,    /* geometry of the later levels as separate launches              */
    AZ_P_LEVEL_LOOP_AT_TZ0   = 32,    /* no one-pass plan for Tz <= 0                                   */
    AZ_P_NO_PAIR_ROWS        = 64,    /* never carry the next level's rows in a pass                    */
    AZ_P_PAIR_ROWS_ALWAYS    = 128,   /* ... at every eligible level                                    */
    AZ_P_NO_WHOLE_TREE       = 256,   /* never the whole-tree / closure pass                            */
    AZ_P_WHOLE_TREE_ALWAYS   = 512,   /* ... whenever the image shape allows                            */
    AZ_P_CLOSURE_ROWS        = 1024,  /* with AZ_P_WHOLE_TREE_ALWAYS: the closure rows                  */
    AZ_P_NO_EARLY_END        = 4096   /* enqueue every level whatever the last search of the shape did  */
};

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
    int32_t spec_rows;                    /* rois forwarded by the speculative pass that serves
                                             levels 1-3 in one launch (0: levels ran one by one) */
    int32_t root_deferred;                /* 1: the root's row rode on level 4's head pass instead
                                             (spec_rows excludes it, level 4 evaluated one more row) */
    int32_t static_plan;                  /* 1: Tz <= 0, all levels went through one head pass of
                                             spec_rows rois (params.reserved bit 5 turns this off)  */
    int32_t n_passes;                     /* head passes (RoIPool -> int6 -> int7 -> heads) the search made */
    int32_t pass_rows[AZ_MAX_LEVELS];     /* rois each of them evaluated (speculative rows included); a search in
                                             its whole-tree form: ONE pass of the image shape's full tree
                                             (pass_rows[0] > spec_rows, static_plan = 0)                   */
    int32_t search_form;                  /* the form the search took (all forms give the same bits): 0 level by level, one
                                             head pass per level (levels 1-3 in one speculative pass); 1 some passes also
                                             carried the next level's rows (pair speculation); 2 ONE pass over the unique rois
                                             of the image shape's full tree; 3 ONE pass over the closure rows (every region any
                                             pruning of the shape's tree can produce); 4 the Tz <= 0 one-pass plan; 5 level by
                                             level in lockstep with the other images of its batch (az_batch_launch: the root and
                                             its children in the first pass, then one pass per level, shared by the batch)      */
    int32_t n_reruns;                     /* times this search had to be run again in another form before it gave this
                                             result (0 normally; e.g. a whole-tree pass over the full tree's rows that lacked
                                             a window the pruned tree needed)                                               */
    int32_t pass_levels[AZ_MAX_LEVELS];   /* per head pass (as pass_rows): bit l set = the pass evaluated the rois of tree level
                                             l + 1 (its own level, the next one's when it carried pair-speculation rows, levels
                                             1-3 for the speculative pass, every level for a whole-tree / one-pass form) --
                                             what a floor that charges ONE weight stream per pass needs (bench.py)           */
} az_stats;

/* ---- lifecycle ----------------------------------------------------------------- */
const char *az_version(void);
/* Layout check for bindings: sizeof(az_params) in the low 16 bits, sizeof(az_stats) in the next 16 (a caller built against
 * another header would hand az_propose_fetch a block of the wrong size -- every fetch clears sizeof(az_stats) bytes -- so a
 * binding compares this with its own structs when it loads the library; lib/aznet_hip/ffi.py does).  Nothing in the reference
 * corresponds: its Cython modules are compiled against their caller. */
int az_abi_sizes(void);
/* Replaces caffe.set_mode_gpu(); caffe.set_device(id) (tools/prop_az.py:88-89). */
int az_create(int device, az_ctx **out);
int az_destroy(az_ctx *ctx);
const char *az_last_error(const az_ctx *ctx);
/* Optional, before az_load_head: per-level region capacity (default 16384) and total
 * candidate capacity (default 16384*11).  Buffers are sized once, for 288 GB of HBM. */
int az_set_limits(az_ctx *ctx, int max_regions, int max_candidates);

/* Optional, before az_load_head.  How int6 (95 % of the head's FLOPs) is evaluated:
 *   0 (default)  fp32 MFMA (v_mfma_f32_32x32x2_f32), bitwise an fmaf chain;
 *   2            fp32 operands as TWO fp16 terms (x * 2^k = x0 + x1, 22 mantissa bits; 2^k brings the largest
 *                |weight| resp. the largest |feature-map value| of the image to [2^14, 2^15), so nothing overflows and
 *                the scaling is exact), three fp16 MFMAs per product (x0*w1 + x1*w0 + x0*w0) with fp32 accumulation:
 *                products good to ~2^-21; the head's outputs are as close to an f64 evaluation as mode 0's
 *                (measured: tests/test_gpu_gemm_modes.py), at 3/16 of the matrix-pipe cost;
 *   3            fp32 operands as THREE bf16 terms (24 mantissa bits: every fp32 value exactly), six bf16 MFMAs per
 *                product (all cross terms of order <= 2), fp32 accumulation: nothing of fp32's precision or range is
 *                given up; 6/16 of the matrix-pipe cost.  (Operands must be finite and below bf16's largest value,
 *                3.39e38: the terms of an inf are (inf, inf - inf), i.e. NaN where fp32 arithmetic gives +-inf.)
 * In modes 2 and 3 every launch, whatever its row count, uses the same per-row arithmetic (a roi's bits do not
 * depend on its batch), and int6 is the only layer that changes.  Any other value is AZ_ERR_INVALID. */
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
/* As _dev, without the closing synchronisation: the transpose is only enqueued on the ctx stream,
 * so handing over the next image's map costs no host round trip.  The source must stay valid (and
 * unmodified) until the next az_propose_fetch / az_propose on this ctx returns. */
int az_set_feature_map_dev_async(az_ctx *ctx, const float *dev_ptr, int C, int H, int W);
/* A map that is already channel-last in HBM ([H][W][C], e.g. a torch.channels_last conv5_3): borrowed as is, no
 * transpose and no copy.  It must stay valid and unmodified while searches use it. */
int az_set_feature_map_dev_nhwc(az_ctx *ctx, const float *dev_ptr, int C, int H, int W);

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
/* az_set_feature_map_dev_async (channels_last = 0: NCHW source) or az_set_feature_map_dev_nhwc (1) followed by
 * az_propose_launch, in one call: one host round trip per image. */
int az_propose_launch_on(az_ctx *ctx, const az_params *p, const float *dev_map, int C, int H, int W,
                         int channels_last);
int az_propose_fetch(az_ctx *ctx, double *boxes_out, float *scores_out, int cap, int *n_out,
                     az_stats *stats);
/* Two lanes (default 1).  With 2, the searches launched through az_propose_launch(_on) take turns between the context's
 * stream and a second stream with per-search buffers of its own (the head's weights are shared): while one image's GEMM holds
 * the matrix cores, the other image's single-workgroup geometry kernels and small head kernels run beside it, so a loop that
 * keeps two searches queued gets consecutive images OVERLAPPED on the GPU (~6 % more images per second at 600x1000).  Results,
 * order of az_propose_fetch (oldest first) and every other call are unchanged; a lane queues up to three searches; the
 * synchronous az_propose, the tuner's variant and variable proposal counts stay on the first lane.  Costs the second lane's
 * buffers (pool5, split-K slabs, geometry: ~1.5 GB at max_regions 4096).  Call with nothing queued.
 * az_next_stream: the hipStream_t the NEXT az_propose_launch(_on) will run on (make it wait for the map's producer there);
 * az_last_stream: the one the most recently launched search runs on (record "search done" events there). */
int az_set_lanes(az_ctx *ctx, int lanes);
void *az_next_stream(az_ctx *ctx);
void *az_last_stream(az_ctx *ctx);
/* A batch of images of ONE shape searched in lockstep -- the images of consecutive iterations of the dataset loop
 * (lib/detect/test.py:508-513), each with its own tree, but every level's rois of ALL of them forwarded in ONE head pass
 * (the reference's roi blob carries Caffe's batch index in column 0, test.py:93-97; there it is always 0).  At a tuned
 * threshold a level of one image is a few dozen rois: passes that are weight streams for one image become matrix work for
 * eight, and the latency chain between two passes (slab sum, int7, heads, geometry kernel) is paid once per level, not once
 * per level and image.  Every image's result is what az_propose gives for it alone, bit for bit (stats: search_form 5).
 *   maps: n device pointers to channel-last maps [H][W][C] (az_set_feature_map_dev_nhwc's layout), valid and unmodified
 *   until the batch's last az_batch_fetch; p: fixed proposal count, not the tuner's variant; 1 <= n <= AZ_BATCH_MAX.
 * The search is level by level for every image (root + its children in the first pass); an image whose tree outgrows a
 * fused kernel's tables, or a batch whose level outgrows max_regions rows, is run again on its own by az_batch_fetch (a
 * batch that would not fit, going by the rows per image of the context's last batch, is enqueued in parts that do).
 * Shapes / settings the lockstep form does not take (fewer than three levels, params.reserved bits 0 / 1 / 4, int6 on the
 * 16-bit matrix cores) are searched one image after the other, same results.  az_batch_fetch returns the images of the
 * OLDEST unfetched batch, i = 0 .. n-1 in order.  Two batches per lane may be in flight (a lane's two run one after the other
 * on its stream: the host enqueues the next while the GPU works on the current one; with az_set_lanes(ctx, 2) four).
 * Uses max_regions-sized geometry buffers per image slot (~25 KB per region), allocated at the first batch. */
int az_batch_launch(az_ctx *ctx, int n, const az_params *p, const float *const *maps_nhwc_dev, int C, int H, int W);
/* The same for images of SEVERAL shapes (a dataset mixes them: VOC has 500x375, 375x500, 500x333 ...): params[b] and the map
 * size H[b] x W[b] are image b's; every image keeps its own pre-pass, RoIPool clamps to its own map and its boxes are clipped
 * to its own size, the head passes are shared as before.  The trees may differ in depth (K of lib/detect/test.py:365-368):
 * the batch runs as many level passes as its deepest tree has, an image's last level gets its final selection where the
 * others go on.  The images of a batch share num_proposals, eps, min_side and the flags, and each has at least three levels
 * (>= 80 px on the short side at MIN_SIDE 10); otherwise they are searched one by one (same results). */
int az_batch_launch_shapes(az_ctx *ctx, int n, const az_params *params, const float *const *maps_nhwc_dev, int C,
                           const int *H, const int *W);
int az_batch_fetch(az_ctx *ctx, int i, double *boxes_out, float *scores_out, int cap, int *n_out, az_stats *stats);
/* All images of the oldest unfetched batch (those not yet fetched one by one) in ONE call: image i's boxes at
 * boxes_out + i * cap * 4, scores at scores_out + i * cap (may be NULL), count in n_out[i] (-1: that image failed), statistics
 * in stats[i] (may be NULL); returns the first error, the other images are collected all the same. */
int az_batch_fetch_all(az_ctx *ctx, double *boxes_out, float *scores_out, int cap, int *n_out, az_stats *stats);
/* the hipStream_t the NEXT az_batch_launch runs on: make it wait for the maps' producers there */
void *az_batch_next_stream(az_ctx *ctx);
/* az_propose_stage_result_dev for the batch launched last (call it right behind az_batch_launch): image i's record to
 * dst_dev + i * pitch_bytes, complete when az_batch_fetch(i) returns -- one strided device-to-device copy for the batch. */
int az_batch_stage_results_dev(az_ctx *ctx, void *dst_dev, size_t pitch_bytes, size_t cap_bytes);
/* Multi-GPU exchange of proposals (image-sharded ranks, one all-gather of fixed-size
 * records; the reference itself is single-process).  A fixed-count search (params.fixed_num) leaves
 * its result in HBM as ONE record of az_result_record_layout(k) bytes: int32 n at n_offset,
 * boxes f64 [k][4] at boxes_offset, scores f32 [k] at scores_offset (rows >= n undefined).
 * az_propose_stage_result_dev, called between az_propose_launch and az_propose_fetch, enqueues a
 * device-to-device copy of that record to dst_dev (e.g. a slot of the RCCL send buffer) on the ctx
 * stream; it is complete when az_propose_fetch returns.  No host hop for the exchanged data. */
int az_result_record_layout(int num_proposals, size_t *bytes, size_t *n_offset, size_t *boxes_offset,
                            size_t *scores_offset);
int az_propose_stage_result_dev(az_ctx *ctx, void *dst_dev, size_t cap_bytes);
/* The exchange itself as ONE ncclAllGather over RCCL / xGMI, no framework in between: every rank
 * contributes `bytes_per_rank` bytes at send_dev (its staged records, padding rows included) and receives all ranks' blocks
 * in rank order at recv_dev (nranks * bytes_per_rank bytes).  It runs on a stream of the context's own
 * (az_comm_stream: a hipStream_t; make it wait for whoever wrote padding rows, make readers of recv_dev wait for it),
 * device-ordered behind everything both lanes have queued when the call is made -- so it may be issued right behind the
 * batch's last az_propose_launch -- and holds neither lane back.  RCCL is bound at run time to the librccl.so the process
 * already holds (PyTorch-ROCm's), else ROCm's.  az_rccl_unique_id: 128 bytes made on rank 0 and handed to every rank by
 * the launcher's own means (a file, torch.distributed's store, MPI); az_rccl_init: collective over the nranks processes,
 * one per GPU.  The reference is single-process: this replaces nothing there. */
int az_rccl_unique_id(void *id_out, size_t cap);
int az_rccl_init(az_ctx *ctx, const void *id, size_t id_bytes, int nranks, int rank);
int az_gather_records(az_ctx *ctx, const void *send_dev, void *recv_dev, size_t bytes_per_rank);
int az_rccl_destroy(az_ctx *ctx);
void *az_comm_stream(az_ctx *ctx);
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
/* The call site itself, apply_nms (lib/detect/test.py:467-484): one nms per class per image, i.e.
 * many small independent problems.  Group g owns dets[offsets[g] .. offsets[g+1]) (rows of 5 f32);
 * keep[offsets[g] ..] receives its kept group-local indices (descending score), n_keep[g] their
 * number.  Groups of <= 256 boxes share one launch (a workgroup each). */
int az_nms_batched(az_ctx *ctx, const float *dets, const int32_t *offsets, int n_groups, double thresh,
                   int64_t *keep, int32_t *n_keep);

/* ---- Fast R-CNN head on the shared conv map (BASELINE config 3) ---------------------- */
/* Replaces caffe.Net(frcnn/test_fc.prototxt, caffemodel) (tools/test_shared.py): the detection
 * head models/Pascal/VGG16/frcnn/test_fc.prototxt:14-145 -- fc6 [n6, C*49], fc7 [n7, n6],
 * cls_score [ncls, n7] (+Softmax), bbox_pred [4*ncls, n7]; Caffe [out, in] layout.  2 <= ncls <= 256 (VOC: 21;
 * COCO, models/COCO/VGG16/frcnn/test_fc.prototxt:97-135: 81). */
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

/* ---- zoom-threshold tuner (lib/detect/tune.py, tools/set_thresh.py) ------------------- */
/* `Bhis` of the tuner's im_propose (tune.py:303, returned at :316) for the last az_propose run
 * with params.reserved bit 2: every region evaluated, level-major, with its zoom score.
 * regions_out [cap,4] f64, zoom_out [cap] f32; either may be NULL. */
int az_last_anchors(az_ctx *ctx, double *regions_out, float *zoom_out, int cap, int *n_out);
/* tune_thresh (tune.py:318-366) keeps the num_images*ANCHORS_PER_IMG largest zoom scores of a
 * whole image set in a heap and returns the smallest of them.  Here the scores stay in HBM:
 * between az_tune_begin and az_tune_end every tuner-variant az_propose appends its anchors' zoom
 * scores to a device pool of `capacity` floats (no host round trip), and az_tune_kth_largest
 * radix-selects the k-th largest: -inf when the pool holds <= k scores, exactly as the heap
 * never overflowing leaves `thresh = -np.inf` (tune.py:326,347-350). */
int az_tune_begin(az_ctx *ctx, long long capacity);
int az_tune_end(az_ctx *ctx);
int az_tune_kth_largest(az_ctx *ctx, long long k, float *value_out, long long *n_total);
/* Multi-GPU merge: every pooled score >= the k-th largest (all of them when the pool holds
 * <= k), unordered; rank 0 pushes the gathered lists into its own pool and selects again. */
int az_tune_top(az_ctx *ctx, long long k, float *scores_out, long long cap, long long *n_out);
int az_tune_push(az_ctx *ctx, const float *scores, long long n);

/* ---- recall evaluation (lib/datasets/imdb.py:120-159) ----------------------------------- */
/* utils.cython_bbox.bbox_overlaps(boxes f64[N,4], query_boxes f64[K,4]) -> f64[N,K]
 * (lib/utils/bbox.pyx:132-172). */
int az_bbox_overlaps(az_ctx *ctx, const double *boxes, int N, const double *query, int K,
                     double *overlaps_out);
/* The matching loop of imdb.evaluate_recall (imdb.py:124-147) for n_images images at once:
 * image i owns boxes[box_off[i]:box_off[i+1]] and gt[gt_off[i]:gt_off[i+1]] (f64 [.,4]); per
 * image, repeatedly take the best remaining (box, gt) pair, record its overlap, retire both.
 * gt_overlaps_out [gt_off[n_images]] in image order, then pick order.  Images without boxes
 * must be left out by the caller (imdb.py:128-129); fewer boxes than gt boxes in an image is
 * AZ_ERR_INVALID (the reference's `assert(gt_ovr >= 0)` fires there). */
int az_recall_match(az_ctx *ctx, int n_images, const double *boxes, const int32_t *box_off,
                    const double *gt, const int32_t *gt_off, double *gt_overlaps_out);

/* ---- image front-end (_get_image_blob, lib/detect/test.py:27-59) ------------------------- */
/* uint8 BGR HWC image (host) -> float32 [3, oh, ow] blob: subtract cfg.PIXEL_MEANS, then
 * cv2.resize(fx=fy=scale, INTER_LINEAR) semantics on f32 (half-pixel centres, edge clamp,
 * horizontal pass then vertical pass).  oh/ow must come from az_image_blob_size
 * (cv2's dsize = round-half-even(dim * scale)).  _dev writes to a device pointer (e.g. the
 * torch tensor the backbone reads), _host to a host array. */
int az_image_blob_size(int h, int w, double scale, int *oh, int *ow);
int az_image_blob_host(az_ctx *ctx, const uint8_t *im, int h, int w, const float *means,
                       double scale, float *blob_out, int oh, int ow);
int az_image_blob_dev(az_ctx *ctx, const uint8_t *im, int h, int w, const float *means,
                      double scale, float *blob_dev, int oh, int ow);

/* As az_image_blob_dev, as ONE step of a pipelined harness: the upload and the kernel are only enqueued -- on `stream`
 * (a hipStream_t, e.g. the stream the backbone runs on; NULL: the ctx stream -- the DEFAULT stream, whose handle is also 0, is
 * passed as hipStreamLegacy, (hipStream_t)1) -- and the call returns; `im` is copied to
 * pinned staging before that, so the caller's array may be reused at once.  Whatever is enqueued on `stream` afterwards
 * (the backbone) finds the blob complete; nothing else is synchronised. */
int az_image_blob_dev_on(az_ctx *ctx, const uint8_t *im, int h, int w, const float *means, double scale,
                         float *blob_dev, int oh, int ow, void *stream);

/* ---- backbone epilogues (context-free; `stream`: a hipStream_t, NULL = the default stream) -------------------- */
/* What follows a VGG16 convolution (models/Pascal/VGG16/az-net/test.prototxt:16-384: Convolution with bias, ReLU in
 * place; Pooling MAX 2x2 / 2, Caffe's ceil mode) in ONE pass over the convolution's output, which PyTorch-ROCm produces
 * without the bias (F.conv2d(x, w, None)): y = max(y + bias[c], 0) in place -- channels_last != 0: y is [hw][C] (C % 4 == 0,
 * 16-byte aligned), else [C][hw].  PyTorch's own bias add and ReLU are two element-wise launches over the same bytes;
 * same fp32 operations, same bits. */
int az_bias_relu(void *stream, float *y, const float *bias, int C, long long hw, int channels_last);
/* The same followed by the 2x2 / 2 max-pool: y [H][W][C] (channels_last != 0: C % 4 == 0, 16-byte aligned) or [C][H][W]
 * -> out [ceil(H/2)][ceil(W/2)][C] / [C][ceil(H/2)][ceil(W/2)]; the last window row / column is clipped by the map's edge
 * (ceil mode).  out = max over the window of max(y + bias, 0), computed as max(max(y) + bias, 0): the same bits (rounding
 * is monotonic). */
int az_bias_relu_pool(void *stream, const float *y, const float *bias, float *out, int C, int H, int W, int channels_last);

/* ---- measurement ------------------------------------------------------------------ */
/* HIP-event timing (events on the ctx stream) of the launches made by az_propose /
 * az_head_forward.  mode bits: 1 = time only the fc GEMM launches, 2 = time every launch
 * group, 4 = keep accumulating across calls until read (otherwise each call starts afresh);
 * 8 = the fp32 fc GEMM launches time THEMSELVES instead (first workgroup in to last workgroup out on the GPU's constant
 * 100 MHz clock, written by the kernels: no event pair on the stream -- an event pair costs ~7 us of stream time and, with
 * two lanes, also spans the time a launch waits for the other lane's GEMM to release the CUs); 32768 launches per call of
 * az_set_profiling, which resets them; 16 (with 8, after a call with 8) = forget the spans recorded so far without
 * touching the GPU -- no stream synchronisation, no copy: what to call right in front of a region to be timed;
 * 0 = off.  names_out: `cap` slots of 32 chars. */
int az_set_profiling(az_ctx *ctx, int mode);
int az_last_kernel_times(az_ctx *ctx, char *names_out, float *ms_out, int32_t *level_out,
                         int cap, int *n_out);
/* Replay az_propose's launch sequence as a hipGraph (captured once per parameter set and feature map;
 * every size is read on the device, so the sequence is fixed).  Same results; the GPU time does not
 * change, the host time inside az_propose_launch drops from ~110 us to ~17 us.  Default: the
 * AZ_GRAPH environment variable (off).  Ignored while kernel timing (az_set_profiling) is on. */
int az_set_graphs(az_ctx *ctx, int on);
/* What the context chooses between the forms of a search by (level by level / pair speculation / one whole-tree pass;
 * all give the same bits): the cost in us of ONE head pass (RoIPool + int6 + reduce + int7 + heads) at a few row counts,
 * ascending, linearly interpolated.  By default the context measures the table on its device the first time a search is
 * launched (~10 ms, HIP events); az_set_pass_costs pins it (n >= 2; tests, or a deployment that has measured its boxes),
 * n = 0 returns to measuring.  az_get_pass_costs reads the table in use (n_out = 0: none yet). */
int az_set_pass_costs(az_ctx *ctx, int n, const int32_t *rows, const double *us);
int az_get_pass_costs(az_ctx *ctx, int32_t *rows_out, double *us_out, int cap, int *n_out);
/* What this box sustains, for reading a roofline fraction apart from the box it was measured on: the fp32-input MFMA rate
 * (TFLOP/s) of a ~3 ms register-only v_mfma_f32_32x32x2_f32 loop on every SIMD, pseudo-random operands (the data-sheet
 * peak, 157.3 TFLOP/s, assumes 2.4 GHz; boxes hold 5-10 % less and differ among themselves), and the rate (TB/s, bytes
 * read + written) of a 1 GiB float4 copy through HBM.  ~40 ms; either output may be NULL.  Blocks until done. */
int az_measure_box(az_ctx *ctx, double *mfma_f32_tflops, double *copy_tb_per_s);
/* The HIP stream the ctx launches on (a hipStream_t). */
void *az_stream(az_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* AZNET_HIP_H */
