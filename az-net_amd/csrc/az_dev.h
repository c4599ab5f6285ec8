// az_dev.h -- shared declarations for the gfx950 kernels behind libaznet_hip.so.
// Wave = 64 lanes everywhere in this tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/aznet_hip.h"

#define AZ_WAVE 64
#define AZ_NSUB AZ_NUM_SUBREG

// Device-resident bookkeeping of one search; every kernel of the level loop reads its
// sizes from here, so the host never synchronises inside the loop.
struct AzCounts {
    int P[AZ_MAX_LEVELS + 1];   // regions per level (B.shape[0]); P[0] = 1
    int U[AZ_MAX_LEVELS];       // unique rois forwarded at that level
    int NC[AZ_MAX_LEVELS];      // candidates kept at that level (after the MIN_SIDE filter)
    int PZ[AZ_MAX_LEVELS];      // regions selected for zoom (len(indZ))
    int CH[AZ_MAX_LEVELS];      // children before _sift_dup
    int ytot[AZ_MAX_LEVELS + 1];// running number of candidates before each level
    int nsel;                   // proposals selected at the end
    int err;                    // bit 0: region capacity, bit 1: candidate capacity, bit 2: children
    int scratch[6];
    // speculative evaluation of levels 1-3 (az_search.hip): |B1|, children of all of B1, rows forwarded
    int specP1, specCH, specU, specPad;
    // tuner's search (lib/detect/tune.py:256-316): rows of the anchor history written so far
    int nhis, hisPad[3];
    // head passes of the fused level loop (az_search.hip): PR[l] = rois the pass launched at level l evaluates (its own
    // unique rois, then -- "pair speculation" -- SPN[l] rows for ALL children of all of its regions, a superset of level
    // l+1's rois, starting at row SPB[l], then the deferred root's row if it rides here); 0 = no pass at that level
    int PR[AZ_MAX_LEVELS], SPB[AZ_MAX_LEVELS], SPN[AZ_MAX_LEVELS];
};

// Geometry of one launch of the head on `U` rois (all device pointers).
struct AzHeadDims {
    int C, H, W;        // feature map
    int pooled;         // 7
    int K6;             // C * pooled * pooled
    int n6, n71, n72;   // fc sizes
    int n7;             // n71 + n72
};

// ----------------------------------------------------------------------------------------
// Box decode shared by the head epilogue and the unit entry point:
// _bbox_pred + _clip_boxes (lib/detect/test.py:106-151).  f64, one rounding per operation;
// exp is evaluated in f32 (np.exp of the f32 deltas) and widened.
static __device__ __forceinline__ void az_decode_box(const double *anchor, const float *d4, int im_h, int im_w, double eps,
                              double *out4)
{
    const double w = anchor[2] - anchor[0] + eps;
    const double h = anchor[3] - anchor[1] + eps;
    const double cx = anchor[0] + 0.5 * w;
    const double cy = anchor[1] + 0.5 * h;
    const double pcx = (double)d4[0] * w + cx;
    const double pcy = (double)d4[1] * h + cy;
    const double pw = (double)expf(d4[2]) * w;
    const double ph = (double)expf(d4[3]) * h;
    double x1 = pcx - 0.5 * pw, y1 = pcy - 0.5 * ph;
    double x2 = pcx + 0.5 * pw, y2 = pcy + 0.5 * ph;
    x1 = x1 > 0.0 ? x1 : 0.0;                       // np.maximum(x1, 0)
    y1 = y1 > 0.0 ? y1 : 0.0;
    const double xm = (double)(im_w - 1), ym = (double)(im_h - 1);
    x2 = x2 < xm ? x2 : xm;                         // np.minimum(x2, W - 1)
    y2 = y2 < ym ? y2 : ym;
    out4[0] = x1; out4[1] = y1; out4[2] = x2; out4[3] = y2;
}


// Candidate filter of _unwrap_adj_pred (lib/detect/test.py:181-185).
static __device__ __forceinline__ bool cand_keep(const double *bx, double min_side)
{
    const double h = bx[3] - bx[1] + 1;
    const double w = bx[2] - bx[0] + 1;
    const double side = (h < w) ? h : w;          // np.minimum(heights, widths)
    return side >= min_side;
}

// Order-preserving map float -> uint (ascending): the sort key of the final selection.
static __device__ __forceinline__ unsigned score_key(float f)
{
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// A kernel argument block read from device memory at a workgroup-uniform address (the batched geometry kernels: image
// blockIdx.y's block of an array the host uploaded before the launch): through the constant address space, so that the
// fields are scalar loads into SGPRs like by-value kernel arguments -- through a generic pointer the compiler keeps every
// field in VGPRs (k_level_geom_b: 128 VGPRs + spills against 98).
#if defined(__HIP_DEVICE_COMPILE__)
#define AZ_UNIFORM_ARGS(T, dst, ptr) T dst; __builtin_memcpy(&dst, (const __attribute__((address_space(4))) T *)(uintptr_t)(ptr), sizeof(T))
#else
#define AZ_UNIFORM_ARGS(T, dst, ptr) T dst = *(ptr)
#endif

// A launch's own span on the constant 100 MHz clock (s_memrealtime): thread 0 of every workgroup folds its entry time
// into ts[0] (min) and its exit time into ts[1] (max); ts == nullptr: nothing.  The host initialises a slot to (~0, 0).
struct AzSpan {
    unsigned long long *ts;
    __device__ __forceinline__ explicit AzSpan(unsigned long long *t) : ts(t)
    {
        if (ts && threadIdx.x == 0) atomicMin(&ts[0], (unsigned long long)wall_clock64());
    }
    __device__ __forceinline__ ~AzSpan()
    {
        if (ts && threadIdx.x == 0) atomicMax(&ts[1], (unsigned long long)wall_clock64());
    }
};

// ---- launchers (az_geom.hip) -----------------------------------------------------------
void azk_init_root(hipStream_t s, AzCounts *cnt, double *B0, int im_h, int im_w);
void azk_rois_keys(hipStream_t s, const double *B, const int *Pptr, int cap, double scale, float dedup,
                   int batch, float *rois, long long *key, int *grp);
void azk_rois_dedup(hipStream_t s, const double *B, const int *Pptr, int cap, double scale, float dedup, int batch,
                    float *rois, long long *key, int *grp, unsigned char *first, int *index, int *inv, float *urois,
                    double *ubox, int *Uptr);
void azk_dedup_rois(hipStream_t s, const long long *key, const int *grp, const int *Nptr, int cap,
                    unsigned char *first, const float *rois, const double *B, int *index, int *inv,
                    float *urois, double *ubox, int *Uptr);
void azk_flags_compact(hipStream_t s, AzCounts *cnt, int level, int capR, int capCand,
                       const double *B, const int *inv, const double *pred_u, const float *score_u,
                       const float *zoom_u, double Tz, double min_side, int force_root,
                       unsigned char *cflag, unsigned char *zflag, int *bc_c, int *bc_z,
                       double *Yall, float *Sall, double *Z, int *zr);
// divide_region children of Z[0..*PZptr) (+ optional provenance ids, see az_geom.hip)
void azk_divide(hipStream_t s, const int *PZptr, int *CHptr, int *err, int capR, int capCh, const double *Z,
                double min_side, int *choff, double *child, long long *ckey, const int *src_off, const int *zr,
                const int *src_base, int src_add, int *csrc);
void azk_dedup_regions(hipStream_t s, const long long *key, const int *Nptr, int cap, int capOut,
                       unsigned char *first, const double *child, double *Bnext, int *Pnext, int *err,
                       const int *csrc, int *srcnext);
void azk_spec_rois(hipStream_t s, const double *root, const double *B1, const double *C2, AzCounts *cnt, int capR,
                   double scale, float *urois);
void azk_spec_lookup(hipStream_t s, int level, const int *Uptr, const int *index, const int *src2,
                     const double *ubox, const float *zoom_s, const float *score_s, const float *delta_s, int im_h,
                     int im_w, double eps, float *zoom_u, float *score_u, float *delta_u, double *pred_u);
// decode + flags for the unit entry point az_decode_filter
void azk_region_keys(hipStream_t s, const double *regions, const int *Nptr, int cap, double min_side,
                     long long *ckey);
void azk_decode_unit(hipStream_t s, const double *anchors, const float *deltas, const float *scores,
                     int R, int im_h, int im_w, double eps, double *pred_u, float *score_u);

// ---- launchers (az_static.hip): the whole tree in one head pass when the zoom test cannot fail (Tz <= 0) ------
struct AzStaticArgs {
    AzCounts *cnt;
    const int *reg_u;             // row of the head pass that serves each region, regions of all levels level-major
    const int *cand_src;          // reg_u[c / 11] * 11 + c % 11 for every candidate slot c (region-major x 11)
    const unsigned *key_u;        // [row][11]: selection key of the decoded box, 0 = dropped by the MIN_SIDE filter
                                  // (written by the tail kernel)
    const double *pred_u;
    const float *score_u, *zoom_u;
    double *Yall;
    float *Sall;
    double Tz;
    int nlev, Utot, capCand;
    int k;                        // (k_static_select) proposals wanted; Yout / Sout = the result block
    double *Yout;
    float *Sout;
    int roff[AZ_MAX_LEVELS + 1];  // first region of each level in reg_u; roff[nlev] = regions of the tree
    int U[AZ_MAX_LEVELS], CH[AZ_MAX_LEVELS];
};
// The level loop's last level (fixed proposal count): append its kept candidates, write its counters and make the final
// top-k in ONE launch (instead of k_flags, k_compact, k_rank_count, k_rank_scatter).
struct AzFinalArgs {
    AzCounts *cnt;
    int level;                    // the last level
    const int *inv;               // region -> unique row of that level
    const unsigned *key_u;        // selection keys of the level's decoded boxes (tail kernel), 0 = dropped
    const double *pred_u;
    const float *score_u, *zoom_u;
    double *Yall;                 // candidates of the earlier levels, [0, ytot[level])
    float *Sall;
    double Tz;
    int force_root, capCand, k;
    double *Yout;
    float *Sout;
};
void azk_final_select(hipStream_t s, const AzFinalArgs &a);
void azk_final_select_batch(hipStream_t s, const AzFinalArgs *args_dev, int n);      // image b: args_dev[b], workgroups (*, b)
void azk_plan_rows(hipStream_t s, const int *inv, const int *Pptr, int capR, int roff, int uoff, int *reg_u);
void azk_plan_cands(hipStream_t s, const int *reg_u, int Rtot, int *cand_src);
// whole-tree speculation: window table over a plan's rows (root_row gets the marker row), map of the speculative rows
void azk_full_tab_build(hipStream_t s, const float *urois, int n_rows, int root_row, float ss, unsigned long long *tab,
                        unsigned T, int *err);
void azk_full_lookup(hipStream_t s, const int *Uptr, const float *urois, const double *ubox, const unsigned long long *tab,
                     unsigned T, int root_row, float ss, const float *delta_all, const float *score_all, const float *zoom_all,
                     int im_h, int im_w, double eps, double min_side, double *pred_v, float *score_v, float *zoom_v,
                     unsigned char *keep_v, unsigned *key_v, int *err);
void azk_full_map(hipStream_t s, const float *spec_urois, int n_spec, float ss, const unsigned long long *tab, unsigned T,
                  int base_extra, int cap_rows, float *urois_full, double *ubox_full, int *map, int *n_extra, int *err);
// closure rows (az_static.hip): rois of n regions; window owners -> rows of the pass + table relabelled to those rows
void azk_closure_rois(hipStream_t s, const double *regs, int n, double scale, float *out);
void azk_closure_compact(hipStream_t s, const float *all, int N, float ss, unsigned long long *tab, unsigned T, int *newrow,
                         float *urois_full, double *ubox_full, int *n_rows, int *err);
void azk_static_candidates(hipStream_t s, const AzStaticArgs &a);
// candidates + final top-k in one launch (fixed proposal count); false: the tree is too large for it
bool azk_static_select(hipStream_t s, const AzStaticArgs &a);

// ---- launchers (az_fused.hip): the first levels inside one workgroup ---------------------
struct AzFusedArgs {
    AzCounts *cnt;
    double *B[2];
    int *srcB[2];
    int *index, *inv, *zr, *choff, *csrc;
    float *rois, *urois;          // (next_dedup) roi projection + dedup of the first level after the fused ones
    int next_dedup, defer_root;
    int cut_next;                 // the host enqueued nothing for the level after the fused ones (it expects the tree to
                                  // end here): regions for it set err bit 1024 and the search is run again without the cut
    int cut_short;                // ... and expects it to end before the third level: the speculative pass evaluated the root
                                  // and its children only; a third level sets err bit 1024
    int spec_next;                // (next_dedup) that level's head pass also carries rows for all children of its regions
    int *choff_next, *crow;       // (spec_next) first child of every region in the all-children list; child -> spec row
    float spatial_scale;
    const double *specB1;         // the pre-pass's B1 (children of the root after _sift_dup)
    int reset, specP1, specCH, specU;   // reset: clear the counters here and restore the (cached) pre-pass's
    const int *choff_all;
    double *ubox, *pred_u, *Yall, *Z, *child;
    float *zoom_u, *score_u, *delta_u, *Sall;
    const float *zoom_s, *score_s, *delta_s;
    double scale, Tz, min_side, eps;
    float dedup;
    int batch, im_h, im_w, nlev, n_fused, capR, capCh, capCand;
    // whole-tree speculation (stab != nullptr): zoom_s / score_s / delta_s are the outputs of the whole-tree pass, row i of
    // the speculative layout is row row_map[i] there, and the first level after the fused ones gets its outputs by window
    // lookup here (into the *_v arrays) instead of a head pass
    const int *row_map; int root_row;
    const unsigned long long *stab; unsigned stabT;
    double *pred_v; float *score_v, *zoom_v; unsigned char *keep_v; unsigned *key_v;
};

// (also clears the counters and writes the root region: it is the first kernel of a fused search)
void azk_spec_prepass(hipStream_t s, AzCounts *cnt, double *root, double *B1, double *child, int *choff_all,
                      float *urois, double scale, double min_side, int capR, int capCh, int im_h, int im_w,
                      int defer_root);
void azk_spec_levels(hipStream_t s, const AzFusedArgs &a);
void azk_spec_levels_batch(hipStream_t s, const AzFusedArgs *args_dev, int n);

// ---- launcher (az_level.hip): one level's geometry (and the final selection) in one workgroup ------
struct AzLevelArgs {
    AzCounts *cnt;
    int level, nlev;
    int cut_next;                  // as AzFusedArgs::cut_next, for the level after this one
    const double *B;               // regions of this level
    double *Bnext;                 // regions of the next level
    const double *pred_u;          // head outputs of this level's unique rois
    const float *score_u, *zoom_u;
    float *urois;                  // next level: unique rois
    int *index;                    // the next level's index on exit
    const int *inv;                // this level's inv_index
    int *inv_next;                 // the next level's inv_index (another buffer: the candidate-copy workgroup reads `inv`
                                   // at its own pace, whenever it is dispatched)
    const unsigned char *keep_u;   // MIN_SIDE filter of this level's decoded boxes (tail kernel)
    const int *Uptr;               // unique rois of this level
    int root_row;                  // 1: the LAST row of this level's head pass (cnt->PR[level] - 1) is the deferred root (az_fused.hip)
    // pair speculation (az_search.hip): this level's head pass also evaluated rows for all children of its regions
    // (lookup_next) -> level l+1's outputs are looked up and decoded here into the *_v arrays instead of a head pass;
    // or the NEXT level's pass shall carry such rows (spec_next): they are appended behind its unique rois here
    int lookup_next, spec_next;    // lookup_next = 2: by RoIPool window in the whole-tree pass (stab), not among pair rows
    const unsigned long long *stab; unsigned stabT; int root_row_full;
    const float *score_all, *zoom_all;   // (lookup_next = 2) outputs of the whole-tree pass (delta_u: its raw deltas)
    const float *delta_u;          // raw box deltas of this level's pass (lookup_next)
    const int *choff_all;          // (lookup_next) written by the previous geometry kernel; (spec_next) written here
    int *choff_next, *crow;        // all-children offsets of the next level's regions; child -> spec row (read / written)
    double *pred_v; float *score_v, *zoom_v; unsigned char *keep_v; unsigned *key_v;
    int im_h, im_w; double eps; float spatial_scale;
    double *ubox;                  // next level: anchor boxes of the unique rois
    double *Yall; float *Sall;     // candidates of the whole search
    double scale, Tz, min_side;
    float dedup;
    int batch, capR, capCh, capCand, force_root;
};
void azk_level_geom(hipStream_t s, const AzLevelArgs &a);
void azk_level_geom_batch(hipStream_t s, const AzLevelArgs *args_dev, int n);

// ---- launchers (az_batch.hip): several images of one shape searched in lockstep, every level's rois in ONE head pass ----
struct AzGatherArgs {
    int n, capR;                              // images in the batch; rows the head's buffers hold
    const int *rows[AZ_BATCH_MAX];            // rows image b forwards in this pass (its counters, or a shape constant)
    int *err[AZ_BATCH_MAX];                   // its search's error word: nonzero = the image forwards nothing more (NULL: not looked at)
    const float *rois[AZ_BATCH_MAX];          // its rois [rows][5]
    const double *ubox[AZ_BATCH_MAX];         // its anchor boxes [rows][4] (NULL: the pass decodes nothing)
    const float *feat[AZ_BATCH_MAX];          // its channel-last map
    int fh[AZ_BATCH_MAX], fw[AZ_BATCH_MAX];   // ... of fh x fw cells
    int im_h[AZ_BATCH_MAX], im_w[AZ_BATCH_MAX];   // the image's size (what its decoded boxes are clipped to)
    int *feat_hw_out, *row_hw_out;            // device tables: [n][2] map sizes for RoIPool, [rows][2] image sizes for the heads
    int *off_out;                             // [AZ_BATCH_MAX + 2]: first row of every image, off_out[n] = off_out[AZ_BATCH_MAX + 1] = rows of the pass
    float *rois_cat; double *ubox_cat;        // the pass's rois (column 0 = image index: Caffe's roi_batch_ind) and anchors
    const float **feats_out;                  // device table RoIPool reads the maps from
};
struct AzScatterArgs {
    int n;
    const int *off;                           // AzGatherArgs::off_out of the pass
    const float *zoom, *score; const double *pred; const unsigned char *keep; const unsigned *key;   // the head's outputs by row (key may be NULL)
    float *zoom_d[AZ_BATCH_MAX], *score_d[AZ_BATCH_MAX]; double *pred_d[AZ_BATCH_MAX];
    unsigned char *keep_d[AZ_BATCH_MAX]; unsigned *key_d[AZ_BATCH_MAX];
};
void azk_batch_gather(hipStream_t s, const AzGatherArgs &a);
void azk_batch_scatter(hipStream_t s, const AzScatterArgs &a);

// ---- launchers (az_head.hip) -----------------------------------------------------------
// feat_nhwc: the conv map transposed to [H][W][C] (azk_nchw_to_nhwc, once per image);
// pool5 comes out bin-major: [roi][ph*7+pw][c]
// With parts > 0, launches of >= min_strips 32-row strips write `parts` planes of 16-bit terms (tile-major,
// azk_act_plane_index; xscale != nullptr: fp16 terms of x * xscale[0], else bf16 round-off terms) instead of fp32
// pool5, for the GEMM on the 16-bit matrix cores (az_head_terms.hip).
void azk_roi_pool(hipStream_t s, const float *feat_nhwc, AzHeadDims d, float spatial_scale,
                  const float *urois, const int *Uptr, int capU, float *pool5, unsigned short *planes,
                  size_t plane_stride, int parts, int min_strips, int coop_tail = 0, const float *xscale = nullptr,
                  const float *const *feats = nullptr,      // feats (device table): the map of roi row r is feats[(int)roi[0]]
                  const int *feat_hw = nullptr);            // ... of feat_hw[2 b] x feat_hw[2 b + 1] cells (NULL: d.H x d.W all)
void azk_nchw_to_nhwc(hipStream_t s, const float *in, float *out, int C, int HW);
// rows [R][C*49]: Caffe order (c*49+p) <-> the bin-major order (p*C+c) pool5 / W6 use in HBM
void azk_permute_k(hipStream_t s, const float *in, float *out, long long rows, int C, int to_bin_major);
// y[M,N] = act(x[M,K] . W[N,K]^T + b) with a fixed S-way split of K (see az_head.hip):
// _gemm writes the S partial slabs part[s][m][n], _reduce adds them in order + bias (+ReLU).
// (launches with more than max_strips 32-row strips are skipped: another kernel owns them)
// W: the layer's [N][K] weights in the tile-major layout of azk_tile_weights (azk_tiled_elems(N, K) floats);
// ldw = K
size_t azk_tiled_elems(int N, int K);
void azk_tile_weights(hipStream_t s, const float *rowmajor, float *tiled, int N, int K);
// many-row shape (az_head12.hip): 12-wave workgroups, one weight tile per <= 12 row strips; same bits as azk_fc_gemm.
// min_rows > 0: the kernel leaves at once when *Mptr is smaller (a launch whose row count only the device knows is
// sent to both kernels, azk_fc_gemm with max_strips = (min_rows - 1) / 32).
int azk_fc_gemm12_prepare();      // per device, with that device current; != 0: keep azk_fc_gemm for every launch
// ts (may be NULL): two words the launch folds its own span into (AzSpan)
void azk_fc_gemm12(hipStream_t s, const float *x, int ldx, const float *W, int ldw, const int *Mptr, int capM, int N,
                   int K, int S, int Kc, float *part, int min_rows = 0, unsigned long long *ts = nullptr);
void azk_fc_gemm(hipStream_t s, const float *x, int ldx, const float *W, int ldw, const int *Mptr, int capM,
                 int N, int K, int S, float *part, int max_strips = 1 << 30, unsigned long long *ts = nullptr);
int azk_fc_chunk(int K, int S);
int azk_gemm_grid();
// ---- int6 on the 16-bit matrix cores (az_head_terms.hip): operands as `parts` planes of 16-bit terms ----
// Activation (pool5) planes: tile-major -- block (row / 32, k / 32) holds 32 rows x 32 terms (2 KB), K padded to a
// multiple of 32 (the padding is zeroed once, at allocation) -- so that a wave's tile load is 1 KB contiguous.
__host__ __device__ inline size_t azk_act_plane_index(int row, int k, int K)
{
    const int KT = (K + 31) >> 5;
    // (inside a block the four 16-byte vectors of a row are XOR-swizzled the way the LDS tile is read -- vector v of
    //  row r sits at v ^ ((r >> 2) & 3) --, so global -> LDS is a linear copy)
    return ((size_t)(row >> 5) * KT + (k >> 5)) * 1024 + (size_t)(row & 31) * 32 + ((((k & 31) >> 3) ^ ((row >> 2) & 3)) << 3) + (k & 7);
}
__host__ __device__ inline size_t azk_act_plane_elems(int rows, int K)
{
    return (size_t)((rows + 31) >> 5) * ((K + 31) >> 5) * 1024;
}
// Weight planes: tile-major (128 rows x 32 terms per 8 KB block, same swizzle), zero-padded; elements per plane.
// scale == 0: bf16 round-off terms; otherwise fp16 terms of w * scale (a power of two).
size_t azk_weight_plane_elems(int N, int K);
void azk_split_weight_planes(hipStream_t s, const float *in, unsigned short *out, int N, int K, int parts, float scale);
// two-term (fp16) mode: scales[0] = power-of-two scale of pool5 for this map, scales[1] = 1 / (scales[0] * sw);
// scales[2..3] are scratch words that must start at zero
void azk_feat_scale(hipStream_t s, const float *feat, long long n, float *scales, float sw);
// part[s][m][n] = chunk s of X . W^T from the planes (parts = 2: fp16 terms, 3 MFMAs per product; 3: bf16 terms, 6);
// same K chunks / slabs as azk_fc_gemm; xplane / wplane: elements per plane
int azk_fc_terms_prepare(int parts);     // per device (dynamic-LDS opt-in), with that device current; != 0: unavailable
int azk_fc_gemm_terms(hipStream_t s, const unsigned short *Xp, int ldx, size_t xplane, const unsigned short *Wp,
                      int ldw, size_t wplane, const int *Mptr, int capM, int N, int K, int S, int Kc, float *part,
                      int parts, const float *scales);
void azk_fc_reduce(hipStream_t s, const float *part, const float *bias, const int *Mptr, int capM, int N,
                   int S, float *y, int ldy, int relu);
// int7_1|int7_2's slab sum + bias + ReLU, then adj_score + adj_bbox + zoom_score (56 outputs) + bias +
// sigmoid + box decode/clip, in one launch
// (vector-ALU products, see az_head.hip).  WtT is the stacked weight block k-major, zero-padded:
// [azk_tail_weight_rows(n71+n72)][64].
#define AZK_TAIL_SPLIT 8      /* k-chunks of the Fast R-CNN head's cls_score|bbox_pred GEMM */
// keep_u (may be NULL): per decoded box, the MIN_SIDE filter of _unwrap_adj_pred (test.py:181-185)
void azk_tail(hipStream_t s, const float *part7, int S7, const float *b7, int n7, const float *WtT, const float *bt,
              const double *ubox, const int *Uptr, int capU, int im_h, int im_w, double eps, float *zoom_u,
              float *score_u, float *delta_u, double *pred_u, unsigned char *keep_u = nullptr, double min_side = 0.0,
              unsigned *key_u = nullptr, const int *row_hw = nullptr);   // row_hw [row][2]: the image every row clips against
size_t azk_tail_lds_bytes(int n7);
size_t azk_tail_weight_rows(int n7);
int azk_fc_split(int K);
// Fast R-CNN head: cls_score | bbox_pred come from one azk_fc_gemm over [5*ncls, n7] weights;
// softmax + per-class box decode per unique roi, then the un-dedup gather.
void azk_det_epilogue(hipStream_t s, const float *part, int S, int ncls, const float *bt, const double *ubox,
                      const int *Uptr, int capU, int im_h, int im_w, double eps, float *prob_u, float *delta_u,
                      double *pred_u);
void azk_det_gather(hipStream_t s, const int *Pptr, const int *inv, int ncls, const float *prob_u,
                    const double *pred_u, float *prob, double *pred);

// ---- launchers (az_select.hip) ---------------------------------------------------------
// Top-k by score (descending, ties: lower index first).  With Yall/Sall/Yout/Sout non-NULL the
// selected boxes/scores are gathered in the same launch.
// rank_scratch: azk_topk_scratch_ints(capN) ints -> the chip-wide counting kernels do the work (NULL:
// the single-workgroup radix select); identical results.
void azk_topk_full(hipStream_t s, const float *scores, const int *Nptr, int capN, int k, int *sel_idx,
                   int *nsel, const double *Yall, const float *Sall, double *Yout, float *Sout, int *rank_scratch);
void azk_topk(hipStream_t s, const float *scores, const int *Nptr, int capN, int k, int *sel_idx,
              int *nsel, int *rank_scratch);
int azk_topk_scratch_ints(int capN);
void azk_thresh_select_full(hipStream_t s, const float *scores, const int *Nptr, int capN, double Tc,
                            int cap_out, int *sel_idx, int *nsel, const double *Yall, const float *Sall,
                            double *Yout, float *Sout);
void azk_gather_sel(hipStream_t s, const int *sel_idx, const int *nsel, int cap, const double *Yall,
                    const float *Sall, double *Yout, float *Sout);
// seq != 0 (keep / nkeep in host-mapped memory that the host polls): every keep entry is (seq << 32) | index and the count
// word is 64 bits, (seq << 32) | n_kept -- the host accepts a word only when it carries the call's sequence number
// removed: n ints of scratch that are zero on entry (and left zero): the chip-wide rank counts
size_t azk_nms_scan_lds_bytes(int n);       // dynamic LDS of the scan kernel (bounds n)
size_t azk_nms_band_words(int n);           // words of `band` for up to n boxes
void azk_nms(hipStream_t s, const float *dets, int n, double thresh, int *order, float *sdets,
             unsigned long long *mask, unsigned long long *band, unsigned long long *removed, long long *keep, int *nkeep,
             unsigned seq = 0);
// many groups of <= azk_nms_small_max() boxes in one launch: group g = dets[goff[g] .. goff[g+1]); gsel lists
// the groups to process; keep[goff[g] ..] gets the kept group-local indices, nkeep[g] their number
void azk_nms_small(hipStream_t s, const float *dets, const int *goff, const int *gsel, int n_sel, double thresh,
                   long long *keep, int *nkeep, int *done_cnt = nullptr, int *done_flag = nullptr, int done_seq = 0,
                   unsigned seq = 0);     // seq != 0: keep entries (seq << 32) | index, counts (seq << 9) | n_kept
int azk_nms_small_max();
// one problem of n <= azk_nms_small_max() boxes in one launch (dets / keep / nkeep may be host-mapped)
void azk_nms_one_small(hipStream_t s, const float *dets, int n, double thresh, long long *keep, int *nkeep, unsigned seq = 0);
#define AZ_TOPK_MAX 4096

// ---- launchers (az_eval.hip): front-end, recall evaluation, threshold tuner ----------------
void azk_image_blob(hipStream_t s, const unsigned char *im, int h, int w, const float *means, double inv_sx,
                    double inv_sy, int oh, int ow, float *out);
void azk_bbox_overlaps(hipStream_t s, const double *boxes, int N, const double *query, int K, double *out);
void azk_recall_match(hipStream_t s, int n_img, const double *boxes, const int *box_off, const double *gt,
                      const int *gt_off, const long long *ov_off, double *ov, double *gt_ovr, int *bad);
void azk_record_anchors(hipStream_t s, const AzCounts *cnt, int level, int capR, int capHis, const double *B,
                        const int *inv, const float *zoom_u, double *hisB, float *hisZ, int *nhis, int *err);
void azk_pool_append(hipStream_t s, const float *src, const int *nptr, int cap_src, float *pool,
                     unsigned long long *pool_n, long long cap);
void azk_pool_hist(hipStream_t s, const float *pool, long long n, unsigned int prefix, int shift,
                   unsigned long long *hist);
void azk_pool_keep(hipStream_t s, const float *pool, long long n, unsigned int kmin, float *dst,
                   unsigned long long *ndst);

// ---- launcher (az_box.hip): what this box sustains (register-only fp32 MFMA loop on all SIMDs; float4 copy) ----------
int azk_measure_box(hipStream_t s, double *mfma_tflops, double *copy_tbps, size_t copy_bytes);

// ---- az_rccl.hip: RCCL bound at run time (the process's own librccl.so); 0 = ok, else *why ---------------------------------
#include <string>
int azk_rccl_unique_id(void *id128, std::string *why);
int azk_rccl_init(const void *id128, int nranks, int rank, void **comm_out, std::string *why);
int azk_rccl_all_gather(void *comm, hipStream_t s, const void *send, void *recv, size_t bytes, std::string *why);
void azk_rccl_destroy(void *comm);
