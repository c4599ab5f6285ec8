/*
 * az_oracle.c -- CPU restatement of the native / Caffe-resident pieces of
 * AZ-Net's proposal search.  TEST INFRASTRUCTURE ONLY: nothing under
 * az-net_amd/ may link, load or call this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / the timed CPU baseline -- never as the product path.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * the upstream az-net tree).  Build: see oracle/Makefile (plain gcc,
 * -ffp-contract=off so f32/f64 expressions round exactly like the
 * reference's C/NumPy arithmetic).
 *
 * Parity status (also in DESIGN.md):
 *   orc_divide_region, orc_sift_dup, orc_nms, orc_bbox_overlaps
 *       pinned by golden vectors generated from the reference's own Cython
 *       (oracle/gen_golden.py -> tests/golden/).
 *   orc_roi_pool, orc_fc, orc_sigmoid
 *       PARITY UNPINNED: the reference delegates these to the caffe-fast-rcnn
 *       submodule (.gitmodules:1-3), which is an empty directory in the
 *       mount with an unknown pin.  They restate the published Fast R-CNN
 *       Caffe layer semantics selected by
 *       models/Pascal/VGG16/az-net/test_fc.prototxt:14-232.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

/* ------------------------------------------------------------------ */
/* divide_region: lib/utils/div.pyx:15-76 (children only, no dedup).   */
/* regions: [P,4] f64 (x1,y1,x2,y2).  out: caller buffer of cap rows.  */
/* Returns number of children written, or -1 when cap is too small.    */
/* ------------------------------------------------------------------ */
long orc_divide_children(const double *regions, long P, double *out, long cap)
{
    long n = 0;
    for (long i = 0; i < P; ++i) {
        const double *r = regions + 4 * i;
        double lengths[2];
        lengths[0] = r[2] - r[0] + 1.0;               /* div.pyx:32 */
        lengths[1] = r[3] - r[1] + 1.0;               /* div.pyx:33 */
        /* np.argmin: first minimum, so a tie picks index 0 (width). */
        int min_ind = (lengths[1] < lengths[0]) ? 1 : 0;   /* div.pyx:35 */
        int max_ind = 1 - min_ind;
        double l_short = lengths[min_ind] / 2;        /* div.pyx:40 */
        /* cdef unsigned int num_long = int(double): truncation. */
        unsigned int num_long = (unsigned int)(lengths[max_ind] / l_short); /* :42 */
        double l_long = lengths[max_ind] / num_long;  /* div.pyx:43 */
        long nblocks = 2L * num_long + (long)(num_long - 1);   /* div.pyx:45 */
        if (n + nblocks > cap) return -1;
        double *sub = out + 4 * n;
        for (unsigned int k = 0; k < 2; ++k)          /* div.pyx:47-56 */
            for (unsigned int j = 0; j < num_long; ++j) {
                double *s = sub + 4 * (k * num_long + j);
                if (min_ind == 0) {
                    s[0] = k * l_short;       s[1] = j * l_long;
                    s[2] = (k + 1) * l_short; s[3] = (j + 1) * l_long;
                } else {
                    s[0] = j * l_long;        s[1] = k * l_short;
                    s[2] = (j + 1) * l_long;  s[3] = (k + 1) * l_short;
                }
            }
        long offset = 2L * num_long;                  /* div.pyx:57 */
        double h_short = l_short / 2;                 /* div.pyx:58 */
        double h_long = l_long / 2;                   /* div.pyx:59 */
        for (unsigned int j = 0; j + 1 < num_long; ++j) {    /* k = 0 only, :60-69 */
            double *s = sub + 4 * (offset + j);
            if (min_ind == 0) {
                s[0] = 0 * l_short + h_short;       s[1] = j * l_long + h_long;
                s[2] = (0 + 1) * l_short + h_short; s[3] = (j + 1) * l_long + h_long;
            } else {
                s[0] = j * l_long + h_long;         s[1] = 0 * l_short + h_short;
                s[2] = (j + 1) * l_long + h_long;   s[3] = (0 + 1) * l_short + h_short;
            }
        }
        for (long b = 0; b < nblocks; ++b) {          /* div.pyx:71-72 */
            sub[4 * b + 0] += r[0]; sub[4 * b + 2] += r[0];
            sub[4 * b + 1] += r[1]; sub[4 * b + 3] += r[1];
        }
        n += nblocks;
    }
    return n;
}

/* ------------------------------------------------------------------ */
/* _sift_dup: lib/utils/div.pyx:78-89.                                 */
/* hash = rint(regions / min_height) . [1,1e3,1e6,1e9]; np.unique with */
/* return_index => ascending hash, first occurrence of each.           */
/* The f64 dot product is exact (integers < 2^53) so an int64 linear   */
/* combination is the same number.                                     */
/* ------------------------------------------------------------------ */
typedef struct { int64_t key; long idx; } orc_ki;

static int orc_ki_cmp(const void *a, const void *b)
{
    const orc_ki *x = (const orc_ki *)a, *y = (const orc_ki *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

int64_t orc_region_key(const double *r, double min_height)
{
    /* np.round == rint (half to even); default FE_TONEAREST. */
    int64_t a = (int64_t)rint(r[0] / min_height);
    int64_t b = (int64_t)rint(r[1] / min_height);
    int64_t c = (int64_t)rint(r[2] / min_height);
    int64_t d = (int64_t)rint(r[3] / min_height);
    return a + 1000LL * b + 1000000LL * c + 1000000000LL * d;
}

long orc_sift_dup(const double *regions, long C, double min_height,
                  double *out, long *index_out)
{
    if (C == 0) return 0;
    orc_ki *ki = (orc_ki *)malloc(sizeof(orc_ki) * (size_t)C);
    for (long i = 0; i < C; ++i) {
        ki[i].key = orc_region_key(regions + 4 * i, min_height);
        ki[i].idx = i;
    }
    qsort(ki, (size_t)C, sizeof(orc_ki), orc_ki_cmp);
    long n = 0;
    for (long i = 0; i < C; ++i) {
        if (i > 0 && ki[i].key == ki[i - 1].key) continue;
        memcpy(out + 4 * n, regions + 4 * ki[i].idx, 4 * sizeof(double));
        if (index_out) index_out[n] = ki[i].idx;
        ++n;
    }
    free(ki);
    return n;
}

/* divide_region = children + _sift_dup (div.pyx:76). */
long orc_divide_region(const double *regions, long P, double min_height,
                       double *out, long cap)
{
    double *tmp = (double *)malloc(sizeof(double) * 4 * (size_t)(cap > 0 ? cap : 1));
    long c = orc_divide_children(regions, P, tmp, cap);
    if (c < 0) { free(tmp); return -1; }
    long n = orc_sift_dup(tmp, c, min_height, out, NULL);
    free(tmp);
    return n;
}

/* ------------------------------------------------------------------ */
/* nms: lib/utils/nms.pyx:17-68.  dets [N,5] f32.  All box arithmetic  */
/* in f32; the threshold is a Python float, so `ovr >= thresh` compares */
/* in double (nms.pyx:17,65).  order = argsort(scores)[::-1]: the       */
/* caller supplies `order` (NumPy's argsort is the reference's sorter;  */
/* with distinct scores any descending sort gives the same order).      */
/* keep: out [N] original indices in visiting order.  Returns count.    */
/* ------------------------------------------------------------------ */
static inline float orc_fmax(float a, float b) { return a >= b ? a : b; }   /* nms.pyx:11-12 */
static inline float orc_fmin(float a, float b) { return a <= b ? a : b; }   /* nms.pyx:14-15 */

long orc_nms(const float *dets, long N, const long *order, double thresh, long *keep)
{
    float *areas = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    char *suppressed = (char *)calloc((size_t)(N > 0 ? N : 1), 1);
    for (long i = 0; i < N; ++i) {
        const float *d = dets + 5 * i;
        /* NumPy f32 array ops: (x2 - x1 + 1) * (y2 - y1 + 1), each op rounded to f32. */
        float w = d[2] - d[0]; w = w + 1.0f;
        float h = d[3] - d[1]; h = h + 1.0f;
        areas[i] = w * h;                              /* nms.pyx:24 */
    }
    long nk = 0;
    for (long _i = 0; _i < N; ++_i) {
        long i = order[_i];
        if (suppressed[i]) continue;
        keep[nk++] = i;
        float ix1 = dets[5 * i], iy1 = dets[5 * i + 1];
        float ix2 = dets[5 * i + 2], iy2 = dets[5 * i + 3];
        float iarea = areas[i];
        for (long _j = _i + 1; _j < N; ++_j) {
            long j = order[_j];
            if (suppressed[j]) continue;
            float xx1 = orc_fmax(ix1, dets[5 * j]);
            float yy1 = orc_fmax(iy1, dets[5 * j + 1]);
            float xx2 = orc_fmin(ix2, dets[5 * j + 2]);
            float yy2 = orc_fmin(iy2, dets[5 * j + 3]);
            float tw = xx2 - xx1; tw = tw + 1.0f;
            float th = yy2 - yy1; th = th + 1.0f;
            float w = orc_fmax(0.0f, tw);              /* nms.pyx:60 */
            float h = orc_fmax(0.0f, th);              /* nms.pyx:61 */
            float inter = w * h;
            float den = iarea + areas[j]; den = den - inter;
            float ovr = inter / den;                   /* nms.pyx:63 */
            if ((double)ovr >= thresh) suppressed[j] = 1;   /* nms.pyx:64-65 */
        }
    }
    free(areas); free(suppressed);
    return nk;
}

/* ------------------------------------------------------------------ */
/* bbox_overlaps: lib/utils/bbox.pyx:132-172 (f64 IoU matrix [N,K]).   */
/* ------------------------------------------------------------------ */
void orc_bbox_overlaps(const double *boxes, long N, const double *query, long K,
                       double *overlaps)
{
    for (long k = 0; k < K; ++k) {
        const double *q = query + 4 * k;
        double box_area = (q[2] - q[0] + 1) * (q[3] - q[1] + 1);
        for (long n = 0; n < N; ++n) {
            const double *b = boxes + 4 * n;
            double o = 0.0;
            double iw = (b[2] < q[2] ? b[2] : q[2]) - (b[0] > q[0] ? b[0] : q[0]) + 1;
            if (iw > 0) {
                double ih = (b[3] < q[3] ? b[3] : q[3]) - (b[1] > q[1] ? b[1] : q[1]) + 1;
                if (ih > 0) {
                    double ua = (b[2] - b[0] + 1) * (b[3] - b[1] + 1) + box_area - iw * ih;
                    o = iw * ih / ua;
                }
            }
            overlaps[n * K + k] = o;
        }
    }
}

/* ------------------------------------------------------------------ */
/* ROIPooling (max), declared at                                        */
/* models/Pascal/VGG16/az-net/test_fc.prototxt:14-25 (pooled 7x7,       */
/* spatial_scale 0.0625).  PARITY UNPINNED (implementation lives in the */
/* absent caffe-fast-rcnn submodule); restates the published Fast R-CNN */
/* ROIPoolingLayer::Forward_cpu: C round() (half away from zero) of     */
/* coord*scale in f32, roi size clamped to >= 1, f32 bin sizes, floor / */
/* ceil bin edges shifted by the roi start and clamped to the map,      */
/* empty bin -> 0, otherwise max over the window.                       */
/* feat: [C,H,W] f32 (batch index in rois[:,0] must be 0).              */
/* rois: [R,5] f32.  out: [R, C*PH*PW] f32, index c*PH*PW + ph*PW + pw. */
/* ------------------------------------------------------------------ */
void orc_roi_pool(const float *feat, int C, int H, int W,
                  const float *rois, long R, int PH, int PW, float spatial_scale,
                  float *out)
{
    for (long r = 0; r < R; ++r) {
        const float *roi = rois + 5 * r;
        int roi_start_w = (int)roundf(roi[1] * spatial_scale);
        int roi_start_h = (int)roundf(roi[2] * spatial_scale);
        int roi_end_w = (int)roundf(roi[3] * spatial_scale);
        int roi_end_h = (int)roundf(roi[4] * spatial_scale);
        int roi_height = roi_end_h - roi_start_h + 1; if (roi_height < 1) roi_height = 1;
        int roi_width = roi_end_w - roi_start_w + 1;  if (roi_width < 1) roi_width = 1;
        float bin_size_h = (float)roi_height / (float)PH;
        float bin_size_w = (float)roi_width / (float)PW;
        for (int c = 0; c < C; ++c) {
            const float *plane = feat + (size_t)c * H * W;
            for (int ph = 0; ph < PH; ++ph)
                for (int pw = 0; pw < PW; ++pw) {
                    int hstart = (int)floorf((float)ph * bin_size_h);
                    int wstart = (int)floorf((float)pw * bin_size_w);
                    int hend = (int)ceilf((float)(ph + 1) * bin_size_h);
                    int wend = (int)ceilf((float)(pw + 1) * bin_size_w);
                    hstart += roi_start_h; hend += roi_start_h;
                    wstart += roi_start_w; wend += roi_start_w;
                    if (hstart < 0) hstart = 0; if (hstart > H) hstart = H;
                    if (hend < 0) hend = 0;     if (hend > H) hend = H;
                    if (wstart < 0) wstart = 0; if (wstart > W) wstart = W;
                    if (wend < 0) wend = 0;     if (wend > W) wend = W;
                    int is_empty = (hend <= hstart) || (wend <= wstart);
                    float m = is_empty ? 0.0f : -FLT_MAX;
                    for (int h = hstart; h < hend; ++h)
                        for (int w = wstart; w < wend; ++w) {
                            float v = plane[h * W + w];
                            if (v > m) m = v;
                        }
                    out[(size_t)r * C * PH * PW + (size_t)c * PH * PW + ph * PW + pw] = m;
                }
        }
    }
}

/* ------------------------------------------------------------------ */
/* InnerProduct (test_fc.prototxt:26-220): y = x . W^T + b, W row-major */
/* [N,K] (Caffe [num_output, input_dim]).  PARITY UNPINNED.  Plain      */
/* k-ascending f32 accumulation; used to cross-check the BLAS path of   */
/* the NumPy oracle on small cases.  relu != 0 applies max(y, 0)        */
/* (ReLU layers, test_fc.prototxt:51-56 et al.).                        */
/* ------------------------------------------------------------------ */
void orc_fc(const float *x, long M, long K, const float *W, const float *b, long N,
            int relu, float *y)
{
    for (long m = 0; m < M; ++m)
        for (long n = 0; n < N; ++n) {
            float acc = 0.0f;
            const float *xr = x + m * K, *wr = W + n * K;
            for (long k = 0; k < K; ++k) { float p = xr[k] * wr[k]; acc = acc + p; }
            acc = acc + b[n];
            if (relu && acc < 0.0f) acc = 0.0f;
            y[m * N + n] = acc;
        }
}

/* Sigmoid layer (test_fc.prototxt:221-232): 1 / (1 + exp(-x)) in f32. */
void orc_sigmoid(float *x, long n)
{
    for (long i = 0; i < n; ++i) x[i] = 1.0f / (1.0f + expf(-x[i]));
}
