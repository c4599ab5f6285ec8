// az_head.hip -- the Caffe-resident part of the AZ head on gfx950
// (models/Pascal/VGG16/az-net/test_fc.prototxt:14-232): RoIPool 7x7 over the cached conv5_3
// map, the InnerProduct layers as an fp32-MFMA GEMM with a fixed split of K, and the fused
// epilogue (11+44+1 outputs, sigmoid, box decode, clip).
//
// Numerics: v_mfma_f32_32x32x2_f32 is bitwise a k-ordered fmaf chain, so every output row is
// a fixed function of its input row: the K range is always cut into the same S chunks and the
// S partial sums are always added in chunk order, whatever the number of rows in flight.  That
// is what makes az_head_forward (unit call) and az_propose (fused loop) agree bit for bit and
// makes the result independent of cfg.SEAR.BATCH_SIZE chunking.
#include "az_dev.h"
#include <hip/hip_fp16.h>
#include <float.h>
#include <stdlib.h>

namespace {

// ======================================================================================
// RoIPool (ROIPooling layer, test_fc.prototxt:14-25; semantics of Fast R-CNN's
// ROIPoolingLayer: C round() of coord*scale, size >= 1, f32 bin size, floor/ceil edges,
// clamp, empty bin -> 0).
// Layouts are chosen for coalescing, not inherited from Caffe:
//   * the map is kept channel-last in HBM ([H][W][C], transposed once per image), so a wave
//     reads 64 consecutive channels of one cell with a single 256-byte load;
//   * pool5 is written bin-major ([roi][ph*7+pw][c]) so the same wave stores 256 contiguous
//     bytes; int6's weight columns are permuted to the same order when the head is loaded
//     (az_load_head), which leaves every dot product's set of terms unchanged.
// One wave per (roi, bin); it walks the channel chunks, so its loads are independent and stay
// in flight together.  No LDS, no barrier: at 517 rois that is 25k independent waves.
// ======================================================================================
__global__ void __launch_bounds__(256) k_nchw_to_nhwc(const float *__restrict__ in, float *__restrict__ out,
                                                      int C, int HW)
{
    __shared__ float t[32][33];
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        t[i][tx] = (c < C && p < HW) ? in[(size_t)c * HW + p] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        if (c < C && p < HW) out[(size_t)p * C + c] = t[tx][i];
    }
}

// rows [R][C*49]: Caffe order (c*49 + p)  <->  bin-major order (p*C + c)
__global__ void k_permute_k(const float *__restrict__ in, float *__restrict__ out, long long rows, int C,
                            int to_bin_major)
{
    const long long K = (long long)C * 49, total = rows * K;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const long long r = idx / K;
        const int k = (int)(idx - r * K);            // destination index within the row
        int src;
        if (to_bin_major) { const int p = k / C, c = k - p * C; src = c * 49 + p; }
        else              { const int c = k / 49, p = k - c * 49; src = p * C + c; }
        out[idx] = in[r * K + src];
    }
}

// InnerProduct weights [N][K] row-major -> the tile-major layout the GEMM streams: [N/128][K/32][128][32],
// zero-padded.  A workgroup's weight tile of one K-step is then ONE contiguous 16 KB block (in the row-major
// layout it is 128 separate 128-byte pieces 100 KB apart: a DRAM page miss each, and the weight-streaming
// bound launches -- the speculative pass, level 4's second tile -- ran at 4.3 of ~5.5 TB/s).
__global__ void k_tile_weights(const float *__restrict__ in, float *__restrict__ out, int N, int K, long long total)
{
    const int KT = (K + 31) >> 5;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int kk = (int)(idx & 31), row = (int)((idx >> 5) & 127);
        const long long tile = idx >> 12;
        const int kt = (int)(tile % KT), nt = (int)(tile / KT);
        const int n = nt * 128 + row, k = kt * 32 + kk;
        out[idx] = (n < N && k < K) ? in[(size_t)n * K + k] : 0.f;
    }
}

// Caffe ROIPooling (test_fc.prototxt:14-25).  One workgroup (4 waves) per (roi, bin): the bin's
// window cells are dealt round-robin to the waves, a lane reads 4 consecutive channels of a cell
// (float4; the map is channel-last), four cells per wave are in flight together, and the four
// partial maxima meet in LDS.  A whole-image roi (window ~70 cells) therefore costs ~5 dependent
// memory round trips instead of 70; a 2x2 window costs one.  max() is exact, so the split does
// not change a bit of the result.  Launches with many rois take the wave-per-bin path instead.
constexpr int ROI_POOL_COOP_MAX = 96;      // rois per launch up to which windows are large (levels 1-3)

// the 16-bit terms of one pool5 value: bf16 round-off terms, or (xs != 0, two-term mode) fp16 terms of x * xs
__device__ __forceinline__ void write_terms(float x, unsigned short *po, size_t plane_stride, int parts, float xs)
{
    if (xs != 0.f) {
        x *= xs;
        for (int q = 0; q < parts; ++q) {
            const __half h = __float2half_rn(x);
            po[q * plane_stride] = __half_as_ushort(h);
            x -= __half2float(h);
        }
        return;
    }
    for (int q = 0; q < parts; ++q) {
        unsigned b = __float_as_uint(x);
        b += 0x7FFFu + ((b >> 16) & 1u);            // bf16 round to nearest even
        po[q * plane_stride] = (unsigned short)(b >> 16);
        x -= __uint_as_float(b & 0xFFFF0000u);
    }
}

__global__ void __launch_bounds__(256) k_roi_pool(const float *__restrict__ feat, AzHeadDims d,
                                                  float spatial_scale, const float *__restrict__ urois,
                                                  const int *Uptr, float *__restrict__ pool5,
                                                  unsigned short *__restrict__ planes, size_t plane_stride,
                                                  int parts, int min_strips, int coop_tail,
                                                  const float *__restrict__ xscale, const float *const *feats,
                                                  const int *__restrict__ feat_hw)
{
    constexpr int P = 7, PP = 49;                 // pooled_h = pooled_w = 7 (test_fc.prototxt:20-21)
    __shared__ __attribute__((aligned(16))) float spart[4][512];
    const int U = *Uptr;
    // launches that the 16-bit-term GEMM will consume get their terms written here directly
    const bool to_planes = parts > 0 && ((U + 31) >> 5) >= min_strips;
    const float xs = xscale ? xscale[0] : 0.f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (coop_tail: the last `coop_tail` rois of a many-roi launch are large all the same -- the deferred root of
    //  az_fused.hip is the whole image -- and take the cooperative form)
    int u_lo = 0;
    if (U > ROI_POOL_COOP_MAX) {
        // many rois = small windows (a level deep in the tree): one wave per (roi, bin) with all the
        // channel chunks of a cell in flight; the cooperative form below would idle three waves
        const int nwaves = (gridDim.x * blockDim.x) >> 6;
        u_lo = max(U - coop_tail, 0);
        for (int item = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; item < u_lo * PP; item += nwaves) {
            const int u = item / PP, p = item - u * PP;
            const int ph = p / P, pw = p - ph * P;
            const float *roi = urois + 5 * (size_t)u;
            // (a batch of images, az_batch.hip: roi[0] is the image's index in the batch -- Caffe's roi_batch_ind)
            const float *fm = feats ? feats[(int)roi[0]] : feat;
            // (... and, with maps of several sizes in the batch, the size of ITS map)
            const int fH = (feats && feat_hw) ? feat_hw[2 * (int)roi[0]] : d.H, fW = (feats && feat_hw) ? feat_hw[2 * (int)roi[0] + 1] : d.W;
            const int rsw = (int)roundf(roi[1] * spatial_scale);
            const int rsh = (int)roundf(roi[2] * spatial_scale);
            const int rew = (int)roundf(roi[3] * spatial_scale);
            const int reh = (int)roundf(roi[4] * spatial_scale);
            int rh = reh - rsh + 1; rh = rh < 1 ? 1 : rh;
            int rw = rew - rsw + 1; rw = rw < 1 ? 1 : rw;
            const float bh = (float)rh / (float)P;
            const float bw = (float)rw / (float)P;
            int hs = (int)floorf((float)ph * bh) + rsh;
            int he = (int)ceilf((float)(ph + 1) * bh) + rsh;
            int ws = (int)floorf((float)pw * bw) + rsw;
            int we = (int)ceilf((float)(pw + 1) * bw) + rsw;
            hs = min(max(hs, 0), fH); he = min(max(he, 0), fH);
            ws = min(max(ws, 0), fW); we = min(max(we, 0), fW);
            const bool empty = (he <= hs) || (we <= ws);
            float *out = pool5 + (size_t)u * d.K6 + (size_t)p * d.C;
            // up to 8 channel chunks (512 channels) per pass, so 8 independent loads are in flight
            for (int cb = 0; cb < d.C; cb += 8 * 64) {
                float m[8];
    #pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = empty ? 0.0f : -FLT_MAX;
                for (int h = hs; h < he; ++h) {
                    const float *row = fm + ((size_t)h * fW + ws) * d.C + cb + lane;
                    for (int w = ws; w < we; ++w, row += d.C) {
                        float v[8];
    #pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = (cb + 64 * j + lane < d.C) ? row[64 * j] : -FLT_MAX;
    #pragma unroll
                        for (int j = 0; j < 8; ++j) m[j] = v[j] > m[j] ? v[j] : m[j];
                    }
                }
    #pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (cb + 64 * j + lane < d.C) {
                        if (!to_planes) {
                            out[cb + 64 * j + lane] = m[j];
                        } else {
                            write_terms(m[j], planes + azk_act_plane_index(u, p * d.C + cb + 64 * j + lane, d.K6), plane_stride,
                                        parts, xs);
                        }
                    }
            }
        }
        if (u_lo >= U) return;
    }
    for (int item = u_lo * PP + blockIdx.x; item < U * PP; item += gridDim.x) {
        const int u = item / PP, p = item - u * PP;
        const int ph = p / P, pw = p - ph * P;
        const float *roi = urois + 5 * (size_t)u;
        const float *fm = feats ? feats[(int)roi[0]] : feat;
        const int fH = (feats && feat_hw) ? feat_hw[2 * (int)roi[0]] : d.H, fW = (feats && feat_hw) ? feat_hw[2 * (int)roi[0] + 1] : d.W;
        const int rsw = (int)roundf(roi[1] * spatial_scale);
        const int rsh = (int)roundf(roi[2] * spatial_scale);
        const int rew = (int)roundf(roi[3] * spatial_scale);
        const int reh = (int)roundf(roi[4] * spatial_scale);
        int rh = reh - rsh + 1; rh = rh < 1 ? 1 : rh;
        int rw = rew - rsw + 1; rw = rw < 1 ? 1 : rw;
        const float bh = (float)rh / (float)P;
        const float bw = (float)rw / (float)P;
        int hs = (int)floorf((float)ph * bh) + rsh;
        int he = (int)ceilf((float)(ph + 1) * bh) + rsh;
        int ws = (int)floorf((float)pw * bw) + rsw;
        int we = (int)ceilf((float)(pw + 1) * bw) + rsw;
        hs = min(max(hs, 0), fH); he = min(max(he, 0), fH);
        ws = min(max(ws, 0), fW); we = min(max(we, 0), fW);
        const bool empty = (he <= hs) || (we <= ws);
        const int nw = we - ws;
        const int ncell = empty ? 0 : (he - hs) * nw;
        float *out = pool5 + (size_t)u * d.K6 + (size_t)p * d.C;
        for (int cb = 0; cb < d.C; cb += 512) {
            // channel quads of this lane in the two 256-channel halves (clamped: C % 4 == 0)
            const int c0 = min(cb + 4 * lane, d.C - 4), c1 = min(cb + 256 + 4 * lane, d.C - 4);
            float4 m0 = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX), m1 = m0;
            for (int i = wave; i < ncell; i += 16) {
                float4 v0[4], v1[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ii = min(i + 4 * q, ncell - 1);      // a repeated cell cannot change a max
                    const int hh = ii / nw;
                    const float *cell = fm + ((size_t)(hs + hh) * fW + (ws + ii - hh * nw)) * d.C;
                    v0[q] = *reinterpret_cast<const float4 *>(cell + c0);
                    v1[q] = *reinterpret_cast<const float4 *>(cell + c1);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    m0.x = v0[q].x > m0.x ? v0[q].x : m0.x; m0.y = v0[q].y > m0.y ? v0[q].y : m0.y;
                    m0.z = v0[q].z > m0.z ? v0[q].z : m0.z; m0.w = v0[q].w > m0.w ? v0[q].w : m0.w;
                    m1.x = v1[q].x > m1.x ? v1[q].x : m1.x; m1.y = v1[q].y > m1.y ? v1[q].y : m1.y;
                    m1.z = v1[q].z > m1.z ? v1[q].z : m1.z; m1.w = v1[q].w > m1.w ? v1[q].w : m1.w;
                }
            }
            *reinterpret_cast<float4 *>(&spart[wave][4 * lane]) = m0;
            *reinterpret_cast<float4 *>(&spart[wave][256 + 4 * lane]) = m1;
            __syncthreads();
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int cc = tid + 256 * half, c = cb + cc;
                if (c < d.C) {
                    float m = spart[0][cc];
                    m = spart[1][cc] > m ? spart[1][cc] : m;
                    m = spart[2][cc] > m ? spart[2][cc] : m;
                    m = spart[3][cc] > m ? spart[3][cc] : m;
                    if (empty) m = 0.0f;
                    if (!to_planes) {
                        out[c] = m;
                    } else {
                        write_terms(m, planes + azk_act_plane_index(u, p * d.C + c, d.K6), plane_stride, parts, xs);
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ======================================================================================
// InnerProduct as split-K GEMM on the fp32 matrix cores.
//   part[s][m][n] = sum_{k in chunk s} x[m][k] * W[n][k]        (k order fixed, see below)
//   y[m][n]       = act(((part[0] + part[1]) + ...) + b[n])
// Workgroup = 256 threads = 4 waves; tile 128 (M) x 128 (N) x 32 (K-step).  Wave w owns the
// 32-column strip w of the tile and all four 32-row strips: 4 accumulators of
// v_mfma_f32_32x32x2_f32 (64 VGPRs).  Row strips beyond M are skipped wave-uniformly, so a
// ragged last M-tile costs only its live strips.
// Operands are staged global -> VGPR (dwordx4) -> LDS (padded rows, ds_write_b128) with the
// next K-step's loads in flight during the current step's MFMAs; fragments come back with
// ds_read_b128: lane l reads 4 consecutive k of row (l & 31) at k-offset 4*(l >> 5), which
// feeds 4 MFMAs (k-pairs {j, 4+j}).  Within each 8-wide k group the accumulation order is
// therefore k = 0,4,1,5,2,6,3,7 -- fixed, and part of the definition above.
// The fp32 MFMA takes 64 cycles per SIMD, so LDS and issue bandwidth are far from binding;
// what matters is keeping HBM loads in flight (2 workgroups/CU, register-prefetched tiles)
// and keeping the weight panel of one (n, s) group on one XCD's L2 (item -> XCD mapping).
// ======================================================================================
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDT = BK + 4;          // padded LDS row (floats): 144 B, conflict-free b128 reads
constexpr int GEMM_GRID = 512;       // 2 workgroups per CU, multiple of 8 XCDs
// tiles of at least this many strips use the MFMA-dense step schedule (mid-step barrier, one instruction
// per MFMA issue slot); single-strip tiles keep the simpler weight-streaming one.  (2 vs 3: -5 us on the
// 49-row speculative pass, neutral elsewhere.)
#ifndef W_AUX_MAX_NRT
#define W_AUX_MAX_NRT 1
#endif
#ifndef W_AUX_STREAM
#define W_AUX_STREAM 2      /* cache policy bits of the weight loads of single-strip (weight-streaming) tiles */
#endif
#ifndef DENSE_MIN_NRT
#define DENSE_MIN_NRT 1     /* (1 vs 2: 130 rows = 3 strips | 1 strip + half: 241 -> 234 us; neutral elsewhere) */
#endif
static int gemm_grid()
{
    static int g = -1;
    if (g < 0) { const char *e = getenv("AZ_GEMM_GRID"); g = e ? atoi(e) : GEMM_GRID; }
    return g;
}

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// Operand tiles are fetched with buffer loads: one descriptor per operand tile (wave-uniform,
// SGPRs), a per-thread byte offset per row that never changes during the K loop (rows past the
// operand's end are clamped to its last row -- their products land in output rows that are
// never stored), and the K position as the scalar offset.  No VALU address arithmetic in the
// loop, and reads past the end of the allocation return 0 instead of faulting.  Values with k
// past the chunk end (only when K is not a multiple of the chunk) are zeroed later, when the
// registers are written to LDS (zero_tail), NOT here: touching a loaded value right after the
// load would make the compiler wait for it and serialise the prefetch.
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const float *base, size_t elems_left)
{
    const size_t bytes = elems_left * sizeof(float);
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes,
                                             0x00020000);
}

template <int AUX = 0>
__device__ __forceinline__ float4 buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

__device__ __forceinline__ float4 zero_tail(float4 v, bool ok)
{
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
}

// Rows of one launch are cut into ceil(strips / 4) m-tiles whose sizes differ by at most one
// 32-row strip (e.g. 17 strips -> 4,4,3,3,3), so work items have near-equal cost.
__device__ __forceinline__ void mtile_rows(int strips, int mt, int t, int &strip0, int &nstrips)
{
    const int base = strips / mt, rem = strips - base * mt;
    strip0 = t * base + (t < rem ? t : rem);
    nstrips = base + (t < rem ? 1 : 0);
}

// One (m-tile, n-tile, k-chunk) work item with NRT live 32-row strips.
// Software pipeline (per wave, so that ONE wave keeps its SIMD's matrix pipe busy; the second
// resident workgroup is cover, not a requirement):
//   * global loads run two K-steps ahead of their use: tile kt+2 is requested at the start of
//     step kt into the register set that step kt-1 emptied, and is written to LDS during step
//     kt+1, while step kt+1's MFMAs run;
//   * the ds_read_b128 fragment loads of the next 8-wide k group are issued before the current
//     group's MFMAs (sched_barrier fences keep the compiler from clustering all reads first);
//   * one barrier per K-step.
// HALF: the tile ends with a 16-row half strip (rows NRT*32 .. NRT*32+15) for the <= 16 rows a launch
// has past its last full strip.  It runs on v_mfma_f32_16x16x4_f32 -- half the matrix-pipe time of a
// padded 32-row strip -- fed so that its accumulation chain is the 32x32x2 one: per 8-wide k group two
// instructions, k slots (0,4,1,5) then (2,6,3,7) (both instructions are bitwise sequential fmaf
// chains over their k slots: dev/mfma_f32_16x16x4_probe.hip), so a row's bits do not depend on which
// kind of strip it lands in.
// QUART: the tile ends with a QUARTER strip instead -- rows NRT*32 .. NRT*32+7, for the <= 8 rows a launch has past its last
// full strip (the 40-row speculative pass of every pruned tree: 8 + 32 rois, the root deferred).  It runs on
// v_mfma_f32_4x4x1_16b_f32: 16 independent 4 x 4 blocks per instruction, K = 1, arranged here as 2 row blocks x 8 column
// blocks = exactly 8 rows x 32 columns, eight instructions per 8-wide k group in the order 0,4,1,5,2,6,3,7 -- bitwise the
// fmaf chain of the other two shapes (dev/mfma_f32_4x4x1_probe.hip), at a quarter of a padded strip's matrix-pipe time.
template <int NRT, bool HALF, bool QUART = false>
__device__ __forceinline__ void fc_tile(const float *__restrict__ X, int ldx, const float *__restrict__ Wt,
                                        int ldw, int M, int N, int m0, int n0, int k0, int kend,
                                        float *__restrict__ slab, float (*sA)[BM * LDT], float (*sB)[BN * LDT])
{
    static_assert(!(HALF && QUART), "a tile ends with a half strip or a quarter strip, not both");
    static_assert(!(HALF || QUART) || NRT <= 3, "the half / quarter strip lives in the tile's fourth strip of LDS rows");
    static_assert(NRT > 0 || HALF || QUART, "a tile has at least a half strip");
    constexpr int NLA = NRT + ((HALF || QUART) ? 1 : 0);    // activation float4s per thread per K-step
    constexpr int NA = NRT > 0 ? NRT : 1;        // (NRT = 0: the tile IS the half strip; arrays keep one unused slot)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lk = (lane >> 5) * 4;
    const int nk = (kend - k0 + BK - 1) / BK;
    floatx4 acch[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

    floatx16 acc[NA];
#pragma unroll
    for (int r = 0; r < NRT; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

    // global -> register staging: thread t owns float4 f = t + 256*i, row f/8, column (f%8)*4 of
    // the (NRT*32) x 32 activation tile and of the 128 x 32 weight tile.
    const int srow = tid >> 3, sc4 = (tid & 7) * 4;
    // descriptors start at the tile's first row; per-thread row offsets are loop-invariant
    const __amdgpu_buffer_rsrc_t rsA = tile_rsrc(X + (size_t)m0 * ldx, (size_t)(M - m0) * ldx);
    // weights are tile-major (k_tile_weights): the 128 x 32 tile of K-step t of this n-tile is the t-th 16 KB block
    const int KT = (ldw + BK - 1) / BK;                  // ldw = K of the whole layer
    const __amdgpu_buffer_rsrc_t rsB = tile_rsrc(Wt + (size_t)(n0 / BN) * KT * (BN * BK), (size_t)KT * (BN * BK));
    unsigned voA[NLA], voB[4];
#pragma unroll
    for (int i = 0; i < NRT; ++i) voA[i] = (unsigned)((min(srow + 32 * i, M - 1 - m0) * ldx + sc4) * 4);
    // half strip: 16 rows x 8 float4 -- threads t and t + 128 stage the same vector (same value, same place)
    if constexpr (HALF || QUART) voA[NRT] = (unsigned)((min(NRT * 32 + (srow & 15), M - 1 - m0) * ldx + sc4) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) voB[i] = (unsigned)(((srow + 32 * i) * BK + sc4) * 4);
    auto gload = [&](int kt, float4 (&ra)[NLA], float4 (&rb)[4]) {
        const unsigned so = (unsigned)(k0 + kt * BK) * 4u;
        const unsigned sob = (unsigned)(k0 / BK + kt) * (unsigned)(BN * BK * 4);
#pragma unroll
        for (int i = 0; i < NLA; ++i) ra[i] = buf_ld(rsA, voA[i], so);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb[i] = buf_ld<(NRT <= W_AUX_MAX_NRT) ? W_AUX_STREAM : 0>(rsB, voB[i], sob);
    };
    auto lstore = [&](int kt, int buf, const float4 (&ra)[NLA], const float4 (&rb)[4]) {
        const bool ok = (k0 + kt * BK + sc4) < kend;
#pragma unroll
        for (int i = 0; i < NRT; ++i)
            *reinterpret_cast<float4 *>(&sA[buf][(srow + 32 * i) * LDT + sc4]) = zero_tail(ra[i], ok);
        if constexpr (HALF || QUART)
            *reinterpret_cast<float4 *>(&sA[buf][(NRT * 32 + (srow & 15)) * LDT + sc4]) = zero_tail(ra[NRT], ok);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4 *>(&sB[buf][(srow + 32 * i) * LDT + sc4]) = zero_tail(rb[i], ok);
    };
    // half-strip fragments of one 8-wide k group: lane = (row or column lane & 15, k slot lane >> 4);
    // slot s feeds k = {0,4,1,5}[s] to the first instruction and k + 2 to the second
    // quarter-strip fragments: lane = (block lane >> 2 = (row block, column block), t = lane & 3): the 8 k of the group of
    // ITS activation row and of ITS weight row (two 16-byte LDS reads each)
    struct HalfFrag { float a1, a2, b1[2], b2[2]; float4 qa[2], qb[2]; };
    floatx4 accq = {0.f, 0.f, 0.f, 0.f};
    const float *qa_base = &sA[0][(NRT * 32 + (lane >> 5) * 4 + (lane & 3)) * LDT];
    const float *qb_base = &sB[0][(wave * 32 + ((lane >> 2) & 7) * 4 + (lane & 3)) * LDT];
    const int hk = ((lane >> 4) & 1) * 4 + (lane >> 5);
    const float *ha_base = &sA[0][(NRT * 32 + (lane & 15)) * LDT + hk];
    const float *hb_base = &sB[0][(wave * 32 + (lane & 15)) * LDT + hk];
    auto hfrag = [&](int buf, int g8, HalfFrag &f) {
        if constexpr (QUART) {
            const float *pa = qa_base + buf * (BM * LDT) + g8 * 8, *pb = qb_base + buf * (BN * LDT) + g8 * 8;
            f.qa[0] = *reinterpret_cast<const float4 *>(pa); f.qa[1] = *reinterpret_cast<const float4 *>(pa + 4);
            f.qb[0] = *reinterpret_cast<const float4 *>(pb); f.qb[1] = *reinterpret_cast<const float4 *>(pb + 4);
        }
        if constexpr (HALF) {
            const float *pa = ha_base + buf * (BM * LDT) + g8 * 8;
            f.a1 = pa[0]; f.a2 = pa[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float *pb = hb_base + buf * (BN * LDT) + h * 16 * LDT + g8 * 8;
                f.b1[h] = pb[0]; f.b2[h] = pb[2];
            }
        }
    };
    auto hmfma = [&](const HalfFrag &f) {
        if constexpr (QUART) {
            accq = __builtin_amdgcn_mfma_f32_4x4x1f32(f.qa[0].x, f.qb[0].x, accq, 0, 0, 0);      // k = 0
            accq = __builtin_amdgcn_mfma_f32_4x4x1f32(f.qa[1].x, f.qb[1].x, accq, 0, 0, 0);      // 4
            accq = __builtin_amdgcn_mfma_f32_4x4x1f32(f.qa[0].y, f.qb[0].y, accq, 0, 0, 0);      // 1
            accq = __builtin_amdgcn_mfma_f32_4x4x1f32(f.qa[1].y, f.qb[1].y, accq, 0, 0, 0);      // 5
            accq = __builtin_amdgcn_mfma_f32_4x4x1f32(f.qa[0].z, f.qb[0].z, accq, 0, 0, 0);      // 2
            accq = __builtin_amdgcn_mfma_f32_4x4x1f32(f.qa[1].z, f.qb[1].z, accq, 0, 0, 0);      // 6
            accq = __builtin_amdgcn_mfma_f32_4x4x1f32(f.qa[0].w, f.qb[0].w, accq, 0, 0, 0);      // 3
            accq = __builtin_amdgcn_mfma_f32_4x4x1f32(f.qa[1].w, f.qb[1].w, accq, 0, 0, 0);      // 7
        }
        if constexpr (HALF) {
#pragma unroll
            for (int h = 0; h < 2; ++h) acch[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a1, f.b1[h], acch[h], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h) acch[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a2, f.b2[h], acch[h], 0, 0, 0);
        }
    };
    const float *a_base = &sA[0][lrow * LDT + lk];
    const float *b_base = &sB[0][(wave * 32 + lrow) * LDT + lk];
    auto frag = [&](int buf, int g8, float4 (&af)[NA], float4 &bf) {
        if constexpr (NRT > 0) {
            bf = *reinterpret_cast<const float4 *>(b_base + buf * (BN * LDT) + g8 * 8);
#pragma unroll
            for (int r = 0; r < NRT; ++r)
                af[r] = *reinterpret_cast<const float4 *>(a_base + buf * (BM * LDT) + r * 32 * LDT + g8 * 8);
        }
    };
    // 4*NRT MFMAs on one 8-wide k group; k-pairs {j, 4 + j}: accumulation order 0,4,1,5,2,6,3,7
    auto mfma8 = [&](const float4 (&af)[NA], const float4 &bf) {
#pragma unroll
        for (int r = 0; r < NRT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[r].x, bf.x, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NRT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[r].y, bf.y, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NRT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[r].z, bf.z, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NRT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[r].w, bf.w, acc[r], 0, 0, 0);
    };
    static_assert(BK == 32, "the step bodies below are written for four 8-wide k groups");
    float4 ra0[NLA], rb0[4], ra1[NLA], rb1[4];
    gload(0, ra0, rb0);
    gload(nk > 1 ? 1 : 0, ra1, rb1);
    __syncthreads();                 // the previous work item's readers are done with both buffers
    lstore(0, 0, ra0, rb0);
    __syncthreads();

    if constexpr (NRT >= DENSE_MIN_NRT && NRT > 0) {
        // MFMA-bound shape.  The barrier sits in the MIDDLE of the K-step: when a wave reaches it
        // two of its four k groups (32 MFMAs, ~2000 cycles) are still queued with their operands
        // already in registers, so barrier skew and the first fragment reads of the next tile
        // hide behind them and the matrix pipe never drains.  Inside a group the other
        // instructions are interleaved one per MFMA issue slot (sched_group_barrier).
        //   a0/a1 hold the fragments of k groups 0/1 of tile kt on entry.
        float4 a0[NA], a1[NA], b0, b1;
        HalfFrag h0, h1;
        frag(0, 0, a0, b0);
        frag(0, 1, a1, b1);
        hfrag(0, 0, h0);
        hfrag(0, 1, h1);
        // with a half strip every phase has 4 more (short) MFMAs and up to 6 more 4-byte DS reads
        // (a quarter strip: 8 short MFMAs and 4 16-byte DS reads)
        constexpr int HM = HALF ? 4 : (QUART ? 8 : 0), HD = HALF ? 3 : (QUART ? 2 : 0);
        constexpr int REM01_ = 4 * NRT + HM - (NLA + 4) - (NRT + 1) - HD, REM01 = REM01_ > 0 ? REM01_ : 0;
        auto step = [&](int kt, int buf, float4 (&rl_a)[NLA], float4 (&rl_b)[4], const float4 (&rw_a)[NLA],
                        const float4 (&rw_b)[4]) {
            __builtin_amdgcn_sched_barrier(0);
            // group 0 MFMAs | request tile kt+2 | fragments of group 2
            gload(kt + 2 < nk ? kt + 2 : nk - 1, rl_a, rl_b);   // unconditional: exact vmcnt bookkeeping
            float4 a2[NA], b2;
            HalfFrag h2;
            frag(buf, 2, a2, b2);
            hfrag(buf, 2, h2);
            mfma8(a0, b0);
            hmfma(h0);
#pragma unroll
            for (int i = 0; i < NLA + 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
            }
#pragma unroll
            for (int i = 0; i < NRT + 1; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
            }
#pragma unroll
            for (int i = 0; i < HD; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, REM01, 0);
            __builtin_amdgcn_sched_barrier(0);
            // group 1 MFMAs | tile kt+1 -> LDS[buf^1] | fragments of group 3
            lstore(kt + 1, buf ^ 1, rw_a, rw_b);                // (past the last step: a tile nobody reads)
            float4 a3[NA], b3;
            HalfFrag h3;
            frag(buf, 3, a3, b3);
            hfrag(buf, 3, h3);
            mfma8(a1, b1);
            hmfma(h1);
#pragma unroll
            for (int i = 0; i < NLA + 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 1);   // 1 DS write
            }
#pragma unroll
            for (int i = 0; i < NRT + 1; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
            }
#pragma unroll
            for (int i = 0; i < HD; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, REM01, 1);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();         // LDS[buf^1] (tile kt+1) complete; everyone's reads of LDS[buf] issued
            __builtin_amdgcn_sched_barrier(0);
            // group 2 MFMAs | fragments of group 0 of tile kt+1
            frag(buf ^ 1, 0, a0, b0);
            hfrag(buf ^ 1, 0, h0);
            mfma8(a2, b2);
            hmfma(h2);
#pragma unroll
            for (int i = 0; i < NRT + 1; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 2);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 2);
            }
#pragma unroll
            for (int i = 0; i < HD; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 2);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 2);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * NRT + HM - (NRT + 1) - HD, 2);
            __builtin_amdgcn_sched_barrier(0);
            // group 3 MFMAs | fragments of group 1 of tile kt+1
            frag(buf ^ 1, 1, a1, b1);
            hfrag(buf ^ 1, 1, h1);
            mfma8(a3, b3);
            hmfma(h3);
#pragma unroll
            for (int i = 0; i < NRT + 1; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 3);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 3);
            }
#pragma unroll
            for (int i = 0; i < HD; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 3);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 3);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * NRT + HM - (NRT + 1) - HD, 3);
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int kt = 0; kt < nk; kt += 2) {
            step(kt, 0, ra0, rb0, ra1, rb1);
            if (kt + 1 < nk) step(kt + 1, 1, ra1, rb1, ra0, rb0);
        }
    } else {
        // Weight-streaming-bound shape (one strip): what matters is bytes in flight, not MFMA
        // density.  MFMAs on LDS[buf] (tile kt); meanwhile request tile kt+2 and write tile kt+1.
        auto step = [&](int kt, int buf, float4 (&rl_a)[NLA], float4 (&rl_b)[4], const float4 (&rw_a)[NLA],
                        const float4 (&rw_b)[4]) {
            float4 a0[NA], a1[NA], b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
            HalfFrag h0, h1;
            frag(buf, 0, a0, b0);
            hfrag(buf, 0, h0);
            gload(kt + 2 < nk ? kt + 2 : nk - 1, rl_a, rl_b);   // unconditional: exact vmcnt bookkeeping
            __builtin_amdgcn_sched_barrier(0);
            frag(buf, 1, a1, b1);
            hfrag(buf, 1, h1);
            __builtin_amdgcn_sched_barrier(0);
            mfma8(a0, b0);
            hmfma(h0);
            __builtin_amdgcn_sched_barrier(0);
            frag(buf, 2, a0, b0);
            hfrag(buf, 2, h0);
            lstore(kt + 1, buf ^ 1, rw_a, rw_b);                // (past the last step: a tile nobody reads)
            __builtin_amdgcn_sched_barrier(0);
            mfma8(a1, b1);
            hmfma(h1);
            __builtin_amdgcn_sched_barrier(0);
            frag(buf, 3, a1, b1);
            hfrag(buf, 3, h1);
            __builtin_amdgcn_sched_barrier(0);
            mfma8(a0, b0);
            hmfma(h0);
            __builtin_amdgcn_sched_barrier(0);
            mfma8(a1, b1);
            hmfma(h1);
            __syncthreads();
        };
        for (int kt = 0; kt < nk; kt += 2) {
            step(kt, 0, ra0, rb0, ra1, rb1);
            if (kt + 1 < nk) step(kt + 1, 1, ra1, rb1, ra0, rb0);
        }
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5).
    const int col = n0 + wave * 32 + (lane & 31);
    if (col < N) {
#pragma unroll
        for (int r = 0; r < NRT; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + r * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < M) slab[(size_t)row * N + col] = acc[r][e];
            }
    }
    if constexpr (QUART) {
        // C/D layout of the 4x4 MFMA blocks: block = lane >> 2 = (row block lane >> 5, column block (lane >> 2) & 7);
        // VGPR e = row e of the block, lane & 3 = its column
        const int qcol = n0 + wave * 32 + ((lane >> 2) & 7) * 4 + (lane & 3);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = m0 + NRT * 32 + (lane >> 5) * 4 + e;
            if (qcol < N && row < M) slab[(size_t)row * N + qcol] = accq[e];
        }
    }
    if constexpr (HALF) {
        // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + e
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int hcol = n0 + wave * 32 + 16 * h + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + NRT * 32 + 4 * (lane >> 4) + e;
                if (hcol < N && row < M) slab[(size_t)row * N + hcol] = acch[h][e];
            }
        }
    }
}

__global__ void __launch_bounds__(256, 2)
k_fc_splitk(const float *__restrict__ X, int ldx, const float *__restrict__ Wt, int ldw,
            const int *Mptr, int capM, int N, int K, int S, int Kc, float *__restrict__ part, int max_strips,
            int half_enabled, unsigned long long *ts)
{
    __shared__ __attribute__((aligned(16))) float sA[2][BM * LDT];
    __shared__ __attribute__((aligned(16))) float sB[2][BN * LDT];

    const int M = *Mptr;
    if (M <= 0) return;
    if (((M + 31) >> 5) > max_strips) return;   // larger launches of this layer belong to another kernel
    // ts (profiling): the launch's own span on the 100 MHz clock -- first workgroup in (ts[0], min), last workgroup out
    // (ts[1], max) -- which an event pair cannot give when another stream's kernels hold the CUs the launch waits for
    AzSpan span(ts);
    const int nt = (N + BN - 1) / BN;
    const int G = nt * S;                       // (n-tile, k-chunk) groups
    // With at least one group per workgroup, a workgroup owns whole groups and walks their
    // m-tiles itself: every workgroup then does the same number of strip-steps whatever M is
    // (no tail imbalance; the weight panel is re-read mt times, from Infinity Cache / HBM,
    // which the 64-cycle fp32 MFMA hides).  With fewer groups than workgroups (the narrow
    // layers) the (group, m-tile) pairs are spread over workgroups instead.
    const bool mloop = (G >= (int)gridDim.x);
    // m-tiles: up to 4 strips each; a narrow layer cuts the rows finer, down to single strips, as
    // long as all (group, m-tile) items still fit the resident workgroups at once -- e.g. 17 strips
    // x 80 groups: 6 tiles of <= 3 strips (480 items) instead of 5 of <= 4 (400 items on 512 slots)
    auto tiles_for = [&](int st) {
        int t = (st + 3) >> 2;
        if (!mloop) t = max(t, min(st, (int)gridDim.x / G));
        if (!mloop && G * t > (int)gridDim.x) {
            // more items than resident workgroups: the launch takes ceil(items / workgroups) turns of one tile each.  Finer
            // tiles (not below 3 strips: 1- and 2-strip tiles keep the matrix pipe waiting for their short accumulation
            // chains) can make the turns shorter -- 25 strips x 80 groups: 7 tiles of <= 4 strips = 2 turns of 4, 9 tiles
            // of <= 3 = 2 turns of 3 (int7 at 773 rows: 100 -> 85 us).  Who computes which rows, never a row's bits.
            const int grid = (int)gridDim.x;
            int best = t, best_cost = ((G * t + grid - 1) / grid) * ((st + t - 1) / t);
            for (int u = t + 1; u <= st && (st + u - 1) / u >= 3; ++u) {
                const int cost = ((G * u + grid - 1) / grid) * ((st + u - 1) / u);
                if (cost < best_cost) { best = u; best_cost = cost; }
            }
            t = best;
        }
        return t;
    };
    int strips = (M + 31) >> 5;
    int mt = tiles_for(strips);
    // <= 16 rows past the last full strip: the last m-tile takes them as a half strip (16x16x4 MFMA),
    // which needs that tile to have <= 3 full strips (one more tile if it would have 4)
    // (group-owned m loops only: for the narrow layers one longer item per group would set the pace.)
    // If the balanced split would end in a 1- or 2-strip tile (the weight-streaming code path), the
    // strips are dealt front-loaded instead -- e.g. 4 strips + half as (3, 1+half), not (2, 2+half).
    bool half_last = false, front = false;
    if ((half_enabled & 1) && mloop && (M & 31) && (M & 31) <= 16 && (M >> 5) >= 1) {
        const int st2 = M >> 5;
        int mt2 = tiles_for(st2), s0, nrt_last;
        mtile_rows(st2, mt2, mt2 - 1, s0, nrt_last);
        if (nrt_last == 4) {
            ++mt2;
            mtile_rows(st2, mt2, mt2 - 1, s0, nrt_last);
            front = nrt_last < 3 && !(half_enabled & 4);
        }
        strips = st2; mt = mt2; half_last = true;
    }
    // spread items (fewer groups than workgroups) with <= 16 rows past the last full strip: those rows are a half strip in
    // the LAST tile, the tiles counted as if it were a strip -- 48 rows = (1 strip | half strip), 144 = (2 | 2 + half)
    if ((half_enabled & 1) && !mloop && (M & 31) && (M & 31) <= 16 && (M >> 5) >= 1) {
        const int st2 = M >> 5;
        const int mt2 = tiles_for(st2 + 1);
        int s0, nrt_last;
        mtile_rows(st2, mt2, mt2 - 1, s0, nrt_last);
        if (mt2 >= 2 && nrt_last <= 3) { strips = st2; mt = mt2; half_last = true; }
    }
    // (half_enabled bit 4: no quarter strips -- AZ_GEMM_QUART=0, measurements)
    const bool quart_last = half_last && (M & 31) <= 8 && !(half_enabled & 16);
    const int nitems = mloop ? G : G * mt;
    // Two tiles per group on a full grid: workgroups b and b + 8 -- the same XCD, hence the same L2 -- take the two tiles
    // of ONE group at the same time (the group's weight panel comes from memory once); b and b + 256 tend to share a CU
    // and take a first and a second tile.  (Adjacent workgroups sit on different XCDs: dealt item by item the panel was
    // fetched twice.)
    const bool xcd_pairs = !mloop && mt == 2 && nitems == (int)gridDim.x && (nitems & 15) == 0 && (((int)gridDim.x / 2) & 15) == 0;
    // (256 % mt != 0 in general: the rotation that makes tile(b + 256) = tile(b) + mt / 2)
    const int pair_rot = (half_enabled & 8) ? 0 : (((mt >> 1) - (int)(gridDim.x / 2) % mt) % mt + mt) % mt;

    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        int g = mloop ? item : item / mt;
        if (xcd_pairs) g = (item >> 4) * 8 + (item & 7);
        const int ntile = g / S, s = g - ntile * S;
        const int n0 = ntile * BN;
        const int k0 = s * Kc;
        const int kend = min(K, k0 + Kc);
        float *slab = part + (size_t)s * capM * N;
        // (spread items: m-tiles differ by one strip -- front ones larger.  Workgroups b and b + gridDim / 2 tend to
        //  share a CU (two resident per CU): the second half of the grid walks the m-tiles rotated by half a turn, so a
        //  CU gets a larger and a smaller tile rather than two large ones.  AZ_GEMM_PAIR=0: plain order.)
        int t_sp = xcd_pairs ? (((item >> 3) + (item >= (int)gridDim.x / 2 ? 1 : 0)) & 1) : item - g * mt;
        // (a whole group rotates or not -- decided by where its first item falls --, so every m-tile is still visited once)
        if (!mloop && !xcd_pairs && pair_rot && (((g * mt) / ((int)gridDim.x / 2)) & 1)) t_sp = (t_sp + pair_rot) % mt;
        const int t_lo = mloop ? 0 : t_sp, t_hi = mloop ? mt : t_lo + 1;
        for (int mtile = t_lo; mtile < t_hi; ++mtile) {
            int strip0, n_rt;                            // live 32-row strips (workgroup-uniform)
            if (!front) mtile_rows(strips, mt, mtile, strip0, n_rt);
            else if (mtile < mt - 1) mtile_rows(strips - 1, mt - 1, mtile, strip0, n_rt);
            else { strip0 = strips - 1; n_rt = 1; }
            const int m0 = strip0 * 32;
#define FC_TILE(NRT_, HALF_) fc_tile<NRT_, HALF_>(X, ldx, Wt, ldw, M, N, m0, n0, k0, kend, slab, sA, sB)
#define FC_TILE_Q(NRT_) fc_tile<NRT_, false, true>(X, ldx, Wt, ldw, M, N, m0, n0, k0, kend, slab, sA, sB)
            if (quart_last && mtile == mt - 1 && n_rt <= 1) {
                // (<= 8 rows past the last full strip, behind at most one strip -- the 40-row speculative pass of a pruned
                //  tree --: a quarter strip; behind two or three strips the half strip stays: its registers are there)
                if (n_rt == 0) FC_TILE_Q(0); else FC_TILE_Q(1);
            } else if (half_last && mtile == mt - 1) {
                switch (n_rt) {
                case 0: FC_TILE(0, true); break;
                case 1: FC_TILE(1, true); break;
                case 2: FC_TILE(2, true); break;
                default: FC_TILE(3, true); break;
                }
            } else {
                switch (n_rt) {
                case 1: FC_TILE(1, false); break;
                case 2: FC_TILE(2, false); break;
                case 3: FC_TILE(3, false); break;
                default: FC_TILE(4, false); break;
                }
            }
#undef FC_TILE
#undef FC_TILE_Q
        }
    }
}

// y = act(sum_s part[s] + b): partial sums added in chunk order (fixed), then the bias.
#ifndef AZ_REDUCE_NT
#define AZ_REDUCE_NT 1
#endif
#ifndef AZ_REDUCE_BATCH
#define AZ_REDUCE_BATCH 8       /* (16 in flight: 32.8 -> 34.4 us at 670 rows: the kernel runs at memory speed either way) */
#endif
__global__ void __launch_bounds__(256)
k_fc_reduce(const float *__restrict__ part, const float *__restrict__ bias, const int *Mptr, int capM,
            int N, int S, float *__restrict__ y, int ldy, int relu)
{
    const int M = *Mptr;
    const int N4 = N >> 2;                       // N % 4 == 0 (checked by the host)
    const long long total = (long long)M * N4;
    const size_t slab = (size_t)capM * N;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(idx / N4);
        const int n = (int)(idx - (long long)m * N4) * 4;
        const float *p = part + (size_t)m * N + n;
        // slabs in batches of RB independent loads (one memory round trip per batch), added in chunk order
        constexpr int RB = AZ_REDUCE_BATCH;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s0 = 0; s0 < S; s0 += RB) {
            float4 t[RB];
#pragma unroll
            for (int j = 0; j < RB; ++j) {
#if AZ_REDUCE_NT
                // (every slab word is read exactly once: a non-temporal load does not park it in L2 -- 38.0 -> 33.9 us at 688 rows)
                const floatx4 v = __builtin_nontemporal_load(reinterpret_cast<const floatx4 *>(p + (size_t)min(s0 + j, S - 1) * slab));
                t[j] = make_float4(v.x, v.y, v.z, v.w);
#else
                t[j] = *reinterpret_cast<const float4 *>(p + (size_t)min(s0 + j, S - 1) * slab);
#endif
            }
#pragma unroll
            for (int j = 0; j < RB; ++j)
                if (s0 + j < S) {
                    if (s0 + j == 0) a = t[j];
                    else { a.x += t[j].x; a.y += t[j].y; a.z += t[j].z; a.w += t[j].w; }
                }
        }
        const float4 b = *reinterpret_cast<const float4 *>(bias + n);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        if (relu) {
            a.x = a.x > 0.f ? a.x : 0.f; a.y = a.y > 0.f ? a.y : 0.f;
            a.z = a.z > 0.f ? a.z : 0.f; a.w = a.w > 0.f ? a.w : 0.f;
        }
        *reinterpret_cast<float4 *>(y + (size_t)m * ldy + n) = a;
    }
}

// ======================================================================================
// Head tail.  adj_score (11) + adj_bbox (44) read int7_1, zoom_score (1) reads int7_2
// (test_fc.prototxt:146-220): 56 outputs of a [56, n71+n72] weight block (zero where an output
// does not read a column; fmaf(x, 0, acc) == acc exactly).  That is 72 kFLOP-pairs per roi -- far
// too little for a matrix-core launch of its own -- so ONE kernel does the products on the vector
// ALUs and finishes the head: bias, Sigmoid on the 12 scores (test_fc.prototxt:221-232), then
// _bbox_pred + _clip_boxes (lib/detect/test.py:106-151) against the roi's own anchor box.
//   workgroup = 16 waves = TAIL_ROWS rois; the int7 rows are staged in LDS (finishing int7_1|int7_2 on the
//   way: slab sum + bias + ReLU); wave w owns the k
//   range [w*kq, (w+1)*kq), lane o the output o: acc[r] = fmaf(x[r][k], WtT[k][o], acc[r]) for k
//   ascending (WtT is k-major, so a wave's weight read is one coalesced 256 B line per k and the
//   x value is an LDS broadcast); out[r][o] = (((p0 + p1) + p2) + ... + p15) + bias[o].
//   The weight block (327 KB) comes from L2 at ~0.5 us per dependent load: 16 waves x 16 loads in
//   flight per lane, double-buffered in registers, keep that latency off the critical path.
// Rows are independent: a roi's bits do not depend on which other rois share the launch.
// ======================================================================================
constexpr int NOUT = AZ_NSUB * 5 + 1;   // 56
constexpr int TAIL_ROWS = 4;
#ifndef AZ_TAIL_WAVES
#define AZ_TAIL_WAVES 16
#endif
constexpr int TAIL_WAVES = AZ_TAIL_WAVES;
#ifndef AZ_TAIL_KB
#define AZ_TAIL_KB 8        /* (16: the k range of a wave is padded 80 -> 96, tail 12.8 / 16.0 us; 8 or 4: 10.5 / 13.2; 32: 25 / 34) */
#endif

constexpr int TAIL_KB = AZ_TAIL_KB;     // k values per register batch

// Every wave walks the same number of k (a multiple of two register batches): the weight block
// and the LDS rows are zero-padded to TAIL_WAVES * kq, so the loop has no conditional loads
// (a guarded load makes the compiler wait for every earlier one) and fmaf(0, 0, acc) == acc.
static int tail_kq(int n7)
{
    const int per = (n7 + TAIL_WAVES - 1) / TAIL_WAVES;
    return (per + 2 * TAIL_KB - 1) / (2 * TAIL_KB) * (2 * TAIL_KB);
}

__global__ void __launch_bounds__(TAIL_WAVES * 64)
k_tail_fused(const float *__restrict__ part7, int S7, size_t slab7, const float *__restrict__ b7, int n7, int kq,
             const float *__restrict__ WtT, const float *__restrict__ bt, const double *__restrict__ ubox,
             const int *Uptr, int im_h, int im_w, double eps, float *zoom_u, float *score_u, float *delta_u,
             double *pred_u, unsigned char *keep_u, double min_side, unsigned *key_u, const int *__restrict__ row_hw)
{
    extern __shared__ __attribute__((aligned(16))) float tail_lds[];
    const int KP = TAIL_WAVES * kq;
    float *xs = tail_lds;                                  // [TAIL_ROWS][KP]
    float *ps = tail_lds + (size_t)TAIL_ROWS * KP;         // [TAIL_WAVES][TAIL_ROWS][64], then outs [TAIL_ROWS][64]
    const int U = *Uptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = wave * kq;
    auto wload = [&](int k0, float (&w)[TAIL_KB]) {       // rows up to KP + TAIL_KB exist (zero)
        const float *p = WtT + (size_t)k0 * 64 + lane;
#pragma unroll
        for (int j = 0; j < TAIL_KB; ++j) w[j] = p[j * 64];
    };
    for (int u0 = blockIdx.x * TAIL_ROWS; u0 < U; u0 += gridDim.x * TAIL_ROWS) {
        const int nr = min(TAIL_ROWS, U - u0);
        float w0[TAIL_KB], w1[TAIL_KB];
        wload(kb, w0);                                     // in flight while the rows are staged
        __syncthreads();
        for (int i = tid * 4; i < TAIL_ROWS * KP; i += TAIL_WAVES * 64 * 4) {
            const int r = i / KP, k = i - r * KP;          // KP % 4 == 0 and n7 % 4 == 0
            // int7 = ReLU(sum of the K-chunk slabs in chunk order + bias): k_fc_reduce's arithmetic, done
            // here while staging (saves a launch; 8 slabs x 5 KB per roi)
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nr && k < n7) {
                const float *p = part7 + (size_t)(u0 + r) * n7 + k;
                // (batches of 8 independent loads: one memory round trip per batch; added in chunk order)
                for (int s0 = 0; s0 < S7; s0 += 8) {
                    float4 t[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) t[j] = *reinterpret_cast<const float4 *>(p + (size_t)min(s0 + j, S7 - 1) * slab7);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (s0 + j < S7) {
                            if (s0 + j == 0) v = t[j];
                            else { v.x += t[j].x; v.y += t[j].y; v.z += t[j].z; v.w += t[j].w; }
                        }
                }
                const float4 bb = *reinterpret_cast<const float4 *>(b7 + k);
                v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
                v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
            }
            *reinterpret_cast<float4 *>(xs + i) = v;
        }
        __syncthreads();
        float acc[TAIL_ROWS];
#pragma unroll
        for (int r = 0; r < TAIL_ROWS; ++r) acc[r] = 0.f;
        auto fma_batch = [&](int k0, const float (&w)[TAIL_KB]) {
#pragma unroll
            for (int j = 0; j < TAIL_KB; j += 4)
#pragma unroll
                for (int r = 0; r < TAIL_ROWS; ++r) {
                    const float4 x = *reinterpret_cast<const float4 *>(xs + r * KP + k0 + j);
                    acc[r] = fmaf(x.x, w[j + 0], acc[r]);
                    acc[r] = fmaf(x.y, w[j + 1], acc[r]);
                    acc[r] = fmaf(x.z, w[j + 2], acc[r]);
                    acc[r] = fmaf(x.w, w[j + 3], acc[r]);
                }
        };
        for (int k = kb; k < kb + kq; k += 2 * TAIL_KB) {
            wload(k + TAIL_KB, w1);
            fma_batch(k, w0);
            wload(k + 2 * TAIL_KB, w0);                    // past the range on the last turn: never used
            fma_batch(k + TAIL_KB, w1);
        }
#pragma unroll
        for (int r = 0; r < TAIL_ROWS; ++r) ps[(wave * TAIL_ROWS + r) * 64 + lane] = acc[r];
        __syncthreads();
        float v = 0.f;
        if (tid < TAIL_ROWS * 64) {
            const int r = tid >> 6;
            v = ps[r * 64 + lane];
            for (int w = 1; w < TAIL_WAVES; ++w) v += ps[(w * TAIL_ROWS + r) * 64 + lane];
            v += lane < NOUT ? bt[lane] : 0.f;
        }
        __syncthreads();
        if (tid < TAIL_ROWS * 64) ps[tid] = v;              // outs[r][o]
        __syncthreads();
        if (tid < TAIL_ROWS * (AZ_NSUB + 1)) {
            const int r = tid / (AZ_NSUB + 1), t = tid - r * (AZ_NSUB + 1);
            if (r < nr) {
                const int u = u0 + r;
                const float *o = ps + r * 64;
                if (t < AZ_NSUB) {
                    // Caffe Sigmoid: 1. / (1. + exp(-x)) -- f32 exp, double divide, f32 store.
                    const float e = expf(-o[t]);
                    const float sc = (float)(1.0 / (1.0 + (double)e));
                    score_u[(size_t)u * AZ_NSUB + t] = sc;
                    float d4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        d4[q] = o[AZ_NSUB + 4 * t + q];
                        delta_u[(size_t)u * 4 * AZ_NSUB + 4 * t + q] = d4[q];
                    }
                    double *pb = pred_u + ((size_t)u * AZ_NSUB + t) * 4;
                    // (a batch of images of several shapes, az_batch.hip: every row clips against ITS image)
                    az_decode_box(ubox + 4 * (size_t)u, d4, row_hw ? row_hw[2 * (size_t)u] : im_h,
                                  row_hw ? row_hw[2 * (size_t)u + 1] : im_w, eps, pb);
                    if (keep_u) {
                        const bool kp = cand_keep(pb, min_side);
                        keep_u[(size_t)u * AZ_NSUB + t] = kp ? 1 : 0;
                        // selection key of the candidate (az_static.hip): 0 = dropped by the MIN_SIDE filter
                        if (key_u) { const unsigned kk = score_key(sc); key_u[(size_t)u * AZ_NSUB + t] = kp ? (kk ? kk : 1u) : 0u; }
                    }
                } else {
                    const float e = expf(-o[NOUT - 1]);
                    zoom_u[u] = (float)(1.0 / (1.0 + (double)e));
                }
            }
        }
    }
}

// ======================================================================================
// Fast R-CNN head epilogue (models/Pascal/VGG16/frcnn/test_fc.prototxt:91-145, driver
// lib/detect/test.py:259-318): cls_score (ncls) and bbox_pred (4*ncls) are one azk_fc_gemm over
// the stacked [5*ncls, n7] weights; this finishes them per unique roi: slab sum + bias, Softmax
// over the classes (Caffe: subtract the max, exp, sum in channel order, divide -- all f32), and
// _bbox_pred + _clip_boxes of every class box against the roi's own anchor.
// One wave per roi; lane l handles classes l, l + 64, l + 128, l + 192 (ncls <= 256: VOC's 21, COCO's 81 --
// models/COCO/VGG16/frcnn/test_fc.prototxt:97-135).
// ======================================================================================
__global__ void __launch_bounds__(256)
k_det_epilogue(const float *__restrict__ part, int capM, int S, int ncls, const float *__restrict__ bt,
               const double *__restrict__ ubox, const int *Uptr, int im_h, int im_w, double eps,
               float *prob_u, float *delta_u, double *pred_u)
{
    const int U = *Uptr;
    const int NO = 5 * ncls;
    const size_t slab = (size_t)capM * NO;
    const int lane = threadIdx.x & 63;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int nslot = (ncls + 63) >> 6;                    // classes per lane (wave-uniform, <= 4)
    for (int u = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; u < U; u += nwaves) {
        const float *row = part + (size_t)u * NO;
        auto out = [&](int o) {
            float a = row[o];
            for (int s = 1; s < S; ++s) a += row[o + s * slab];
            return a + bt[o];
        };
        float x[4], e[4];
        float m = -FLT_MAX;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = lane + 64 * q;
            x[q] = (q < nslot && c < ncls) ? out(c) : -FLT_MAX;
            m = x[q] > m ? x[q] : m;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const float t = __shfl_xor(m, d, 64); m = t > m ? t : m; }
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = (q < nslot && lane + 64 * q < ncls) ? expf(x[q] - m) : 0.f;
        float sum = 0.f;                                   // channel order, like Caffe's gemv with ones
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < nslot) {
                const int cend = min(64, ncls - 64 * q);
                for (int c = 0; c < cend; ++c) sum += __shfl(e[q], c, 64);
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = lane + 64 * q;
            if (q < nslot && c < ncls) {
                prob_u[(size_t)u * ncls + c] = e[q] / sum;
                float d4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    d4[k] = out(ncls + 4 * c + k);
                    delta_u[(size_t)u * 4 * ncls + 4 * c + k] = d4[k];
                }
                az_decode_box(ubox + 4 * (size_t)u, d4, im_h, im_w, eps, pred_u + ((size_t)u * ncls + c) * 4);
            }
        }
    }
}

// scores / boxes of every input region = those of its unique roi (test.py:310-312)
__global__ void k_det_gather(const int *Pptr, const int *__restrict__ inv, int ncls, const float *__restrict__ prob_u,
                             const double *__restrict__ pred_u, float *prob, double *pred)
{
    const int P = *Pptr;
    const long long total = (long long)P * ncls;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / ncls), c = (int)(i - (long long)r * ncls);
        const size_t src = (size_t)inv[r] * ncls + c;
        prob[i] = prob_u[src];
#pragma unroll
        for (int q = 0; q < 4; ++q) pred[(size_t)i * 4 + q] = pred_u[src * 4 + q];
    }
}

}  // namespace

// --------------------------------------------------------------------------------------
void azk_roi_pool(hipStream_t s, const float *feat_nhwc, AzHeadDims d, float spatial_scale, const float *urois,
                  const int *Uptr, int capU, float *pool5, unsigned short *planes, size_t plane_stride, int parts,
                  int min_strips, int coop_tail, const float *xscale, const float *const *feats, const int *feat_hw)
{
    (void)capU;
    hipLaunchKernelGGL(k_roi_pool, dim3(4096), dim3(256), 0, s, feat_nhwc, d, spatial_scale, urois, Uptr, pool5,
                       planes, plane_stride, parts, min_strips, coop_tail, xscale, feats, feat_hw);
}

void azk_permute_k(hipStream_t s, const float *in, float *out, long long rows, int C, int to_bin_major)
{
    hipLaunchKernelGGL(k_permute_k, dim3(4096), dim3(256), 0, s, in, out, rows, C, to_bin_major);
}

size_t azk_tiled_elems(int N, int K) { return (size_t)((N + BN - 1) / BN) * ((K + BK - 1) / BK) * (BN * BK); }

void azk_tile_weights(hipStream_t s, const float *in, float *out, int N, int K)
{
    hipLaunchKernelGGL(k_tile_weights, dim3(4096), dim3(256), 0, s, in, out, N, K, (long long)azk_tiled_elems(N, K));
}

void azk_nchw_to_nhwc(hipStream_t s, const float *in, float *out, int C, int HW)
{
    hipLaunchKernelGGL(k_nchw_to_nhwc, dim3((HW + 31) / 32, (C + 31) / 32), dim3(256), 0, s, in, out, C, HW);
}

// Fixed number of K chunks per layer (independent of M; see the header comment).
int azk_fc_split(int K)
{
    // (AZ_SPLIT_BIG: measurements -- the number of chunks is part of a row's bits, so it is one per process)
    static int big = -1;
    if (big < 0) { const char *e = getenv("AZ_SPLIT_BIG"); big = (e && atoi(e) > 0) ? atoi(e) : 16; }
    if (K >= 16384) return big;
    // (AZ_SPLIT_MID: measurements, as AZ_SPLIT_BIG -- int7's chunk count)
    static int mid = -1;
    if (mid < 0) { const char *e = getenv("AZ_SPLIT_MID"); mid = (e && atoi(e) > 0) ? atoi(e) : 8; }
    if (K >= 2048) return mid;
    if (K >= 512) return 2;
    return 1;
}

static int fc_chunk(int K, int S)
{
    // chunk length: a multiple of the K-step so that chunk boundaries never depend on M
    int Kc = (K + S - 1) / S;
    return (Kc + BK - 1) / BK * BK;
}

void azk_fc_gemm(hipStream_t s, const float *x, int ldx, const float *W, int ldw, const int *Mptr, int capM,
                 int N, int K, int S, float *part, int max_strips, unsigned long long *ts)
{
    static int half = -1;                  // AZ_GEMM_HALF=0: pad the last rows to a full strip instead (measurements)
    if (half < 0) {
        const char *e = getenv("AZ_GEMM_HALF"), *f = getenv("AZ_GEMM_BALANCED"), *g = getenv("AZ_GEMM_PAIR");     // (bit 2: balanced tiles, bit 3: no pair rotation: measurements)
        const char *q = getenv("AZ_GEMM_QUART");
        half = ((e ? atoi(e) : 1) ? 1 : 0) | ((f && atoi(f)) ? 4 : 0) | ((g && !atoi(g)) ? 8 : 0) | ((q && !atoi(q)) ? 16 : 0);
    }
    hipLaunchKernelGGL(k_fc_splitk, dim3(gemm_grid()), dim3(256), 0, s, x, ldx, W, ldw, Mptr, capM, N, K, S,
                       fc_chunk(K, S), part, max_strips, half, ts);
}

int azk_fc_chunk(int K, int S) { return fc_chunk(K, S); }
int azk_gemm_grid() { return gemm_grid(); }

void azk_fc_reduce(hipStream_t s, const float *part, const float *bias, const int *Mptr, int capM, int N,
                   int S, float *y, int ldy, int relu)
{
    hipLaunchKernelGGL(k_fc_reduce, dim3(1024), dim3(256), 0, s, part, bias, Mptr, capM, N, S, y, ldy, relu);
}

size_t azk_tail_lds_bytes(int n7)
{
    return ((size_t)TAIL_ROWS * TAIL_WAVES * tail_kq(n7) + TAIL_WAVES * TAIL_ROWS * 64) * sizeof(float);
}

// rows of the k-major, zero-padded weight block [rows][64] the tail kernel reads
size_t azk_tail_weight_rows(int n7) { return (size_t)TAIL_WAVES * tail_kq(n7) + 2 * TAIL_KB; }


void azk_tail(hipStream_t s, const float *part7, int S7, const float *b7, int n7, const float *WtT, const float *bt,
              const double *ubox, const int *Uptr, int capU, int im_h, int im_w, double eps, float *zoom_u,
              float *score_u, float *delta_u, double *pred_u, unsigned char *keep_u, double min_side, unsigned *key_u,
              const int *row_hw)
{
    int grid = (capU + TAIL_ROWS - 1) / TAIL_ROWS;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(k_tail_fused, dim3(grid), dim3(TAIL_WAVES * 64), azk_tail_lds_bytes(n7), s, part7, S7,
                       (size_t)capU * n7, b7, n7, tail_kq(n7), WtT, bt, ubox, Uptr, im_h, im_w, eps, zoom_u, score_u,
                       delta_u, pred_u, keep_u, min_side, key_u, row_hw);
}

void azk_det_epilogue(hipStream_t s, const float *part, int S, int ncls, const float *bt, const double *ubox,
                      const int *Uptr, int capU, int im_h, int im_w, double eps, float *prob_u, float *delta_u,
                      double *pred_u)
{
    hipLaunchKernelGGL(k_det_epilogue, dim3(256), dim3(256), 0, s, part, capU, S, ncls, bt, ubox, Uptr, im_h, im_w,
                       eps, prob_u, delta_u, pred_u);
}

void azk_det_gather(hipStream_t s, const int *Pptr, const int *inv, int ncls, const float *prob_u,
                    const double *pred_u, float *prob, double *pred)
{
    hipLaunchKernelGGL(k_det_gather, dim3(256), dim3(256), 0, s, Pptr, inv, ncls, prob_u, pred_u, prob, pred);
}
