// az_geom_dev.h -- device helpers shared by the multi-workgroup geometry kernels (az_geom.hip)
// and the single-workgroup fused kernels of the first levels (az_fused.hip).  All f64/f32
// arithmetic here is written in the reference's operation order and must be compiled with
// -ffp-contract=off.
#pragma once
#include "az_dev.h"

static __device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Exclusive prefix sum of one int per thread across the block (blockDim.x <= 1024).
static __device__ int block_excl_scan(int v, int *total, int *wsum /* >= 17 ints of LDS */)
{
    const int lane = lane_id(), wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    __syncthreads();                       // wsum may still be read from a previous call
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < nw; ++w) { int t = wsum[w]; wsum[w] = run; run += t; }
        wsum[16] = run;
    }
    __syncthreads();
    *total = wsum[16];
    return wsum[wid] + inc - v;
}

// lib/detect/test.py:61-97 (_get_rois_blob: f64 box * scale -> f32) and :212-214 (hash of
// np.round(rois * DEDUP_BOXES) . [1,1e3,1e6,1e9,1e12]; exact integers, so int64 here).
// `r` is the region's position in its level: with cfg.DEDUP_BOXES <= 0 the reference skips the
// dedup (`if cfg.DEDUP_BOXES > 0:`, test.py:211,246 / :281,312), i.e. index = inv_index = identity;
// the position itself is then the key (all distinct, ascending), which gives exactly that.
static __device__ __forceinline__ long long roi_and_key(const double *box, double scale, float dedup, float *roi5, int r)
{
    long long h = 0, mult = 1000;
    roi5[0] = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float x = (float)(box[c] * scale);
        roi5[1 + c] = x;
        const float t = rintf(x * dedup);              // np.round: half to even, in f32
        h += (long long)t * mult;
        mult *= 1000;
    }
    return dedup > 0.0f ? h : (long long)r;
}

// divide_region, lib/utils/div.pyx:15-76.
struct DivPlan { int min_ind; unsigned num_long; double l_short, l_long; };

static __device__ __forceinline__ DivPlan div_plan(const double *r)
{
    DivPlan p;
    const double L0 = r[2] - r[0] + 1.0, L1 = r[3] - r[1] + 1.0;   // div.pyx:32-33
    p.min_ind = (L1 < L0) ? 1 : 0;                                 // np.argmin: tie -> width
    const double Lmin = p.min_ind ? L1 : L0, Lmax = p.min_ind ? L0 : L1;
    p.l_short = Lmin / 2;                                          // div.pyx:40
    const double q = Lmax / p.l_short;
    p.num_long = (p.l_short > 0.0 && q < 1.0e6) ? (unsigned)q : 0u;   // int(): truncation, div.pyx:42
    p.l_long = p.num_long ? Lmax / p.num_long : 0.0;               // div.pyx:43
    return p;
}

static __device__ __forceinline__ int div_nchildren(const DivPlan &p)
{
    return p.num_long ? (int)(3 * p.num_long - 1) : 0;             // div.pyx:45
}

// Child `bi` of parent r (div.pyx:47-72) and its _sift_dup hash (div.pyx:86).
static __device__ __forceinline__ long long div_child(const double *r, const DivPlan &p, int bi, double min_side,
                                                      double *c)
{
    const double h_short = p.l_short / 2, h_long = p.l_long / 2;   // div.pyx:58-59
    double s_lo, s_hi, l_lo, l_hi;          // short-axis / long-axis cell bounds
    if (bi < (int)(2 * p.num_long)) {       // grid cells, index k*num_long + j (div.pyx:47-56)
        const unsigned k = (unsigned)bi / p.num_long, j = (unsigned)bi - k * p.num_long;
        s_lo = k * p.l_short; s_hi = (k + 1) * p.l_short;
        l_lo = j * p.l_long;  l_hi = (j + 1) * p.l_long;
    } else {                                // half-offset cells, k = 0 (div.pyx:60-69)
        const unsigned j = (unsigned)bi - 2 * p.num_long;
        s_lo = 0 * p.l_short + h_short; s_hi = (0 + 1) * p.l_short + h_short;
        l_lo = j * p.l_long + h_long;   l_hi = (j + 1) * p.l_long + h_long;
    }
    if (p.min_ind == 0) { c[0] = s_lo; c[1] = l_lo; c[2] = s_hi; c[3] = l_hi; }
    else                { c[0] = l_lo; c[1] = s_lo; c[2] = l_hi; c[3] = s_hi; }
    c[0] += r[0]; c[2] += r[0]; c[1] += r[1]; c[3] += r[1];        // div.pyx:71-72
    long long h = 0, mult = 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        h += (long long)rint(c[q] / min_side) * mult;              // div.pyx:86
        mult *= 1000;
    }
    return h;
}

// Candidate filter of _unwrap_adj_pred (lib/detect/test.py:181-185).
static __device__ __forceinline__ bool cand_keep(const double *bx, double min_side)
{
    const double h = bx[3] - bx[1] + 1;
    const double w = bx[2] - bx[0] + 1;
    const double side = (h < w) ? h : w;          // np.minimum(heights, widths)
    return side >= min_side;
}
