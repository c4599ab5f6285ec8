// az_eval.hip -- the callers either side of the proposal path:
//   k_image_blob      image front-end: uint8 BGR HWC -> mean-subtracted, bilinearly resized f32 CHW
//                     (_get_image_blob, lib/detect/test.py:27-59; cv2.resize INTER_LINEAR on f32)
//   k_bbox_overlaps   utils.cython_bbox.bbox_overlaps (lib/utils/bbox.pyx:132-172), f64
//   k_recall_match    the greedy box<->ground-truth matching of imdb.evaluate_recall
//                     (lib/datasets/imdb.py:120-147), one workgroup per image
//   k_record_anchors  anchor history `Bhis` of the tuner's search (lib/detect/tune.py:256-316)
//   k_pool_*          k-th largest zoom score over a whole image set (tune_thresh, tune.py:318-366)
// All of it is HBM/latency-bound integer / f64 work: no LDS tiling beyond block reductions.
#include "az_dev.h"

namespace {

// ---------------------------------------------------------------------------------------------
// cv2.resize(src f32, fx=fy=scale, INTER_LINEAR): sample position of destination pixel d is
// (d + 0.5) / scale - 0.5 (double), narrowed to f32, floor -> s, fraction -> a (f32);
// s < 0 -> (0, a=0); s >= n-1 -> (n-1, a=0).  Horizontal pass first (S[s]*(1-a) + S[s+1]*a),
// then vertical, each product and sum rounded to f32 (no fma).
struct Tap { int s0, s1; float w0, w1; };

__device__ __forceinline__ Tap make_tap(int d, double inv_scale, int n)
{
    float f = (float)(((double)d + 0.5) * inv_scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= n - 1) { s = n - 1; f = 0.f; }
    Tap t;
    t.s0 = s;
    t.s1 = s + 1 < n ? s + 1 : n - 1;
    t.w0 = 1.f - f;
    t.w1 = f;
    return t;
}

__global__ void __launch_bounds__(256) k_image_blob(const unsigned char *__restrict__ im, int h, int w, float m0,
                                                     float m1, float m2, double inv_sx, double inv_sy, int oh, int ow,
                                                     float *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= ow) return;
    const Tap tx = make_tap(x, inv_sx, w);
    const Tap ty = make_tap(y, inv_sy, h);
    const float mean[3] = {m0, m1, m2};
    const unsigned char *r0 = im + (size_t)ty.s0 * w * 3;
    const unsigned char *r1 = im + (size_t)ty.s1 * w * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a00 = (float)r0[tx.s0 * 3 + c] - mean[c], a01 = (float)r0[tx.s1 * 3 + c] - mean[c];
        const float a10 = (float)r1[tx.s0 * 3 + c] - mean[c], a11 = (float)r1[tx.s1 * 3 + c] - mean[c];
        const float h0 = a00 * tx.w0 + a01 * tx.w1;
        const float h1 = a10 * tx.w0 + a11 * tx.w1;
        out[((size_t)c * oh + y) * ow + x] = h0 * ty.w0 + h1 * ty.w1;
    }
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double iou_f64(const double *b, const double *q)
{
    const double box_area = (q[2] - q[0] + 1.0) * (q[3] - q[1] + 1.0);
    const double iw = (b[2] < q[2] ? b[2] : q[2]) - (b[0] > q[0] ? b[0] : q[0]) + 1.0;
    if (!(iw > 0.0)) return 0.0;
    const double ih = (b[3] < q[3] ? b[3] : q[3]) - (b[1] > q[1] ? b[1] : q[1]) + 1.0;
    if (!(ih > 0.0)) return 0.0;
    const double ua = (b[2] - b[0] + 1.0) * (b[3] - b[1] + 1.0) + box_area - iw * ih;
    return iw * ih / ua;
}

__global__ void __launch_bounds__(256) k_bbox_overlaps(const double *__restrict__ boxes, int N,
                                                        const double *__restrict__ query, int K,
                                                        double *__restrict__ out)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)N * K) return;
    const int n = (int)(t / K), k = (int)(t % K);
    out[t] = iou_f64(boxes + (size_t)n * 4, query + (size_t)k * 4);
}

// One workgroup per image: overlaps [N,K] into `ov` (global scratch), then K rounds of
//   argmax over columns of the column maxima (first index on ties, as np.argmax), record it,
//   blank that box row and that gt column with -1                       (imdb.py:133-145)
// bad[i] is set when a round finds no non-negative entry (the reference's assert fires).
#define RM_T 256
__global__ void __launch_bounds__(RM_T) k_recall_match(const double *__restrict__ boxes, const int *__restrict__ box_off,
                                                        const double *__restrict__ gt, const int *__restrict__ gt_off,
                                                        const long long *__restrict__ ov_off, double *__restrict__ ov,
                                                        double *__restrict__ gt_ovr, int *__restrict__ bad)
{
    const int img = blockIdx.x, tid = threadIdx.x;
    const int b0 = box_off[img], N = box_off[img + 1] - b0;
    const int g0 = gt_off[img], K = gt_off[img + 1] - g0;
    double *o = ov + ov_off[img];
    for (int t = tid; t < N * K; t += RM_T)
        o[t] = iou_f64(boxes + (size_t)(b0 + t / K) * 4, gt + (size_t)(g0 + t % K) * 4);
    __shared__ double s_val[RM_T];
    __shared__ int s_k[RM_T], s_n[RM_T];
    __syncthreads();
    for (int round = 0; round < K; ++round) {
        double best = -2.0;
        int bk = 0x7fffffff, bn = 0;
        for (int k = tid; k < K; k += RM_T) {
            double cm = o[k];
            int cn = 0;
            for (int n = 1; n < N; ++n) {
                const double v = o[(size_t)n * K + k];
                if (v > cm) { cm = v; cn = n; }
            }
            if (cm > best) { best = cm; bk = k; bn = cn; }       // k ascends per thread: first max kept
        }
        s_val[tid] = best; s_k[tid] = bk; s_n[tid] = bn;
        __syncthreads();
        for (int st = RM_T / 2; st > 0; st >>= 1) {
            if (tid < st) {
                const double v2 = s_val[tid + st];
                const int k2 = s_k[tid + st];
                if (v2 > s_val[tid] || (v2 == s_val[tid] && k2 < s_k[tid])) {
                    s_val[tid] = v2; s_k[tid] = k2; s_n[tid] = s_n[tid + st];
                }
            }
            __syncthreads();
        }
        const double gv = s_val[0];
        const int gk = s_k[0], gn = s_n[0];
        __syncthreads();
        if (tid == 0) {
            gt_ovr[g0 + round] = gv;
            if (!(gv >= 0.0)) bad[img] = 1;
        }
        for (int k = tid; k < K; k += RM_T) o[(size_t)gn * K + k] = -1.0;
        for (int n = tid; n < N; n += RM_T) o[(size_t)n * K + gk] = -1.0;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Bhis rows of level `level`: [B | zoom] for every region evaluated (tune.py:303), appended
// after the rows of the earlier levels.
__global__ void __launch_bounds__(256) k_record_anchors(const AzCounts *__restrict__ cnt, int level, int cap,
                                                         const double *__restrict__ B, const int *__restrict__ inv,
                                                         const float *__restrict__ zoom_u, double *__restrict__ hisB,
                                                         float *__restrict__ hisZ, int *__restrict__ nhis,
                                                         int *__restrict__ err)
{
    int off = 0;
    for (int l = 0; l < level; ++l) off += cnt->P[l];
    const int P = cnt->P[level];
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0) {
        *nhis = off + P;
        if (off + P > cap) atomicOr(err, 16);
    }
    if (r >= P || off + r >= cap) return;
    const double *b = B + (size_t)r * 4;
    double *o = hisB + (size_t)(off + r) * 4;
    o[0] = b[0]; o[1] = b[1]; o[2] = b[2]; o[3] = b[3];
    hisZ[off + r] = zoom_u[inv[r]];
}

// ---------------------------------------------------------------------------------------------
// Score pool: append, then k-th largest by an MSB-first radix select over order-preserving keys.
__device__ __forceinline__ unsigned int fkey(float f)
{
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// pool[*pool_n ...] <- src[0..*nptr); the running size advances in the follow-up 1-thread kernel
// (same stream), so the host never needs the per-image count.  pool_n[1] counts dropped scores.
__global__ void __launch_bounds__(256) k_pool_append(const float *__restrict__ src, const int *__restrict__ nptr,
                                                      float *__restrict__ pool,
                                                      const unsigned long long *__restrict__ pool_n, long long cap)
{
    const int n = *nptr;
    const long long base = (long long)pool_n[0];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && base + i < cap) pool[base + i] = src[i];
}

__global__ void k_pool_advance(const int *__restrict__ nptr, unsigned long long *__restrict__ pool_n, long long cap)
{
    const long long n = *nptr, base = (long long)pool_n[0];
    const long long room = cap - base;
    const long long take = n < room ? n : (room > 0 ? room : 0);
    pool_n[0] = (unsigned long long)(base + take);
    pool_n[1] += (unsigned long long)(n - take);
}

// histogram of byte `shift/8` of the keys whose higher bytes equal `prefix`
__global__ void __launch_bounds__(256) k_pool_hist(const float *__restrict__ pool, long long n, unsigned int prefix,
                                                    int shift, unsigned long long *__restrict__ hist)
{
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned int hi_mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned int k = fkey(pool[i]);
        if ((k & hi_mask) == (prefix & hi_mask)) atomicAdd(&h[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// keep every score whose key is >= `kmin` (stream compaction into `dst`, order not preserved)
__global__ void __launch_bounds__(256) k_pool_keep(const float *__restrict__ pool, long long n, unsigned int kmin,
                                                    float *__restrict__ dst, unsigned long long *__restrict__ ndst)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = pool[i];
        if (fkey(v) >= kmin) dst[atomicAdd(ndst, 1ull)] = v;
    }
}

}  // namespace

void azk_image_blob(hipStream_t s, const unsigned char *im, int h, int w, const float *means, double inv_sx,
                    double inv_sy, int oh, int ow, float *out)
{
    dim3 grid((ow + 255) / 256, oh);
    hipLaunchKernelGGL(k_image_blob, grid, dim3(256), 0, s, im, h, w, means[0], means[1], means[2], inv_sx, inv_sy, oh,
                       ow, out);
}

void azk_bbox_overlaps(hipStream_t s, const double *boxes, int N, const double *query, int K, double *out)
{
    const long long tot = (long long)N * K;
    if (tot <= 0) return;
    hipLaunchKernelGGL(k_bbox_overlaps, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, boxes, N, query, K, out);
}

void azk_recall_match(hipStream_t s, int n_img, const double *boxes, const int *box_off, const double *gt,
                      const int *gt_off, const long long *ov_off, double *ov, double *gt_ovr, int *bad)
{
    if (n_img <= 0) return;
    hipLaunchKernelGGL(k_recall_match, dim3(n_img), dim3(RM_T), 0, s, boxes, box_off, gt, gt_off, ov_off, ov, gt_ovr,
                       bad);
}

void azk_record_anchors(hipStream_t s, const AzCounts *cnt, int level, int capR, int capHis, const double *B,
                        const int *inv, const float *zoom_u, double *hisB, float *hisZ, int *nhis, int *err)
{
    hipLaunchKernelGGL(k_record_anchors, dim3((capR + 255) / 256), dim3(256), 0, s, cnt, level, capHis, B, inv, zoom_u,
                       hisB, hisZ, nhis, err);
}

void azk_pool_append(hipStream_t s, const float *src, const int *nptr, int cap_src, float *pool,
                     unsigned long long *pool_n, long long cap)
{
    hipLaunchKernelGGL(k_pool_append, dim3((cap_src + 255) / 256), dim3(256), 0, s, src, nptr, pool, pool_n, cap);
    hipLaunchKernelGGL(k_pool_advance, dim3(1), dim3(1), 0, s, nptr, pool_n, cap);
}

void azk_pool_hist(hipStream_t s, const float *pool, long long n, unsigned int prefix, int shift,
                   unsigned long long *hist)
{
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_pool_hist, dim3((unsigned)blocks), dim3(256), 0, s, pool, n, prefix, shift, hist);
}

void azk_pool_keep(hipStream_t s, const float *pool, long long n, unsigned int kmin, float *dst,
                   unsigned long long *ndst)
{
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_pool_keep, dim3((unsigned)blocks), dim3(256), 0, s, pool, n, kmin, dst, ndst);
}
