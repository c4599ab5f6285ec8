// az_head_bf16.hip -- the int6 GEMM on the bf16 matrix cores with fp32 operands split into
// bf16 terms ("split-bf16").
//
// gfx950's fp32-input MFMA runs at 1/16 of the bf16 MFMA rate.  An fp32 value is the sum of its
// bf16 round-off terms: x = x0 + x1 (+ x2), each term the bf16 rounding of what the previous
// ones left.  The product of two such sums, accumulated in fp32 by the MFMA, needs
//   PARTS = 2:  x0*w1 + x1*w0 + x0*w0                      (3 MFMAs, products good to ~2^-16)
//   PARTS = 3:  x1*w1 + x0*w2 + x2*w0 + x0*w1 + x1*w0 + x0*w0   (6 MFMAs, ~2^-24: fp32-grade;
//               measured on int6: max error 4e-8 vs 3e-6 for an fp32 BLAS)
// at 3/16 resp. 6/16 of the fp32-MFMA cost.  The terms ("planes") are produced once: for the
// weights when the head is loaded, for pool5 by the RoIPool kernel.
//
// Same fixed K chunks -> slabs -> k_fc_reduce as the fp32 kernel (az_head.hip), so the two are
// interchangeable per launch.  Used for launches with >= 3 row strips (> 64 rows); smaller
// launches are weight-streaming bound and stay on the fp32 kernel (4 bytes/element of weights,
// not 2*PARTS).  With the matrix pipe 5x (PARTS = 2) cheaper, what binds is operand traffic, hence
// the 256-row tiles, 8 waves and one workgroup per CU described below.
#include "az_dev.h"
#include <hip/hip_fp16.h>
#include <stdlib.h>

#ifndef AZ_W_AUX
#define AZ_W_AUX 0          /* buffer-load cache policy of the weight stream (2 = nt measured 25 % slower) */
#endif

namespace {

constexpr int BN = 128, BK = 32;                // BK in bf16 elements = two 32x32x16 MFMA k blocks
constexpr int LDR = BK + 8;                     // padded LDS row (bf16): 80 B, conflict-free b128 reads

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short f2bf(float x)      // round to nearest even
{
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// rows [R][K] fp32 -> PARTS planes [p][R][K] of 16-bit terms (plane stride = rows_cap * K).
// scale == 0: bf16 round-off terms.  scale != 0 (a power of two): fp16 terms of x * scale (|x * scale| < 65504).
__global__ void k_split_planes(const float *__restrict__ in, unsigned short *__restrict__ out, long long n,
                               long long plane_stride, int parts, float scale)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float x = in[i];
        if (scale != 0.f) {
            x *= scale;
            for (int p = 0; p < parts; ++p) {
                const __half h = __float2half_rn(x);
                out[p * plane_stride + i] = __half_as_ushort(h);
                x -= __half2float(h);
            }
            continue;
        }
        for (int p = 0; p < parts; ++p) {
            const unsigned short h = f2bf(x);
            out[p * plane_stride + i] = h;
            x -= bf2f(h);
        }
    }
}

// Scale of the fp16 terms of pool5 (two-term mode): pool5 values are maxima of feature-map values, so
// |pool5| <= max |map|; sc[0] = sx = the power of two that brings that maximum into [2^14, 2^15), sc[1] = 1 / (sx * sw)
// (what the GEMM's epilogue multiplies its sums by; sw = the weights' power-of-two scale).  One launch: block maxima ->
// atomicMax on the bits (|x| >= 0: unsigned order = float order) -> the last block to arrive writes the scales and
// clears the scratch words sc[2], sc[3] for the next call.
__global__ void __launch_bounds__(256) k_feat_scale(const float *__restrict__ feat, long long n, float *sc, float sw)
{
    __shared__ float wm[4];
    float m = 0.f;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4 *>(feat)[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, fabsf(feat[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        unsigned *w = reinterpret_cast<unsigned *>(sc);
        atomicMax(w + 2, __float_as_uint(m));
        __threadfence();
        if (atomicAdd(w + 3, 1u) == gridDim.x - 1) {
            const float mx = __uint_as_float(atomicExch(w + 2, 0u));
            float sx = 1.f;
            if (mx > 0.f && mx < INFINITY) {
                int e;
                (void)frexpf(mx, &e);                    // mx < 2^e
                sx = ldexpf(1.f, 15 - e);
            }
            sc[0] = sx;
            sc[1] = 1.f / (sx * sw);
            atomicExch(w + 3, 0u);
        }
    }
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc16(const unsigned short *base, size_t elems_left)
{
    const size_t bytes = elems_left * sizeof(unsigned short);
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes,
                                             0x00020000);
}

__device__ __forceinline__ void mtile_rows(int strips, int mt, int t, int &strip0, int &nstrips)
{
    const int base = strips / mt, rem = strips - base * mt;
    strip0 = t * base + (t < rem ? t : rem);
    nstrips = base + (t < rem ? 1 : 0);
}

// One (m-tile, n-tile, k-chunk) work item.  Workgroup = 8 waves on a 256 (M) x 128 (N) x 32 (K)
// tile: wave w owns column strip (w & 3) and the row strips of half (w >> 2); the live strips
// of a ragged tile are split evenly between the two halves.  NRTW = strips of THIS wave
// (0..4, wave-uniform); the cooperative parts (global loads, LDS writes, barriers) are the same
// for every wave.  LDS stage layout: A planes [PARTS][256][LDR], then B planes [PARTS][128][LDR].
//   * global loads run two K-steps ahead (two register sets), LDS is double-buffered;
//   * the barrier sits between the two 16-wide k blocks of a step, so when a wave reaches it the
//     second block's MFMAs are still queued with their fragments already in registers.
template <int NRTW, int PARTS, int WAVES, bool F16>
__device__ __forceinline__ void fc_tile_bf16(const unsigned short *__restrict__ Xp, int ldx, size_t xplane,
                                             const unsigned short *__restrict__ Wp, int ldw, size_t wplane,
                                             int M, int N, int m0, int wstrip0, int n0, int k0, int kend,
                                             float *__restrict__ slab, unsigned short *lds, float oscale)
{
    constexpr int BMT = WAVES * 32;                          // rows of the tile: 256 (8 waves) or 128 (4 waves)
    constexpr int RSTEP = WAVES * 16;                        // rows covered by one pass of all threads (4 vectors/row)
    constexpr int NB = BN / RSTEP;                           // B vectors per thread per plane (1 or 2)
    constexpr int ATILE = BMT * LDR, BTILE = BN * LDR;      // one plane of A / of B (elements)
    constexpr int STAGE = PARTS * (ATILE + BTILE);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cstrip = wave & 3;
    const int lrow = lane & 31, lk = (lane >> 5) * 8;
    const int nk = (kend - k0 + BK - 1) / BK;

    floatx16 acc[NRTW > 0 ? NRTW : 1];
#pragma unroll
    for (int r = 0; r < NRTW; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

    // global -> register staging.  A plane tile: BMT rows x 64 B -> thread t owns the 16-byte vectors
    // (row t/4 + i*RSTEP, col t%4), i < 2; B plane tile (128 rows): i < NB.
    const int srow = tid >> 2, sc8 = (tid & 3) * 8;
    __amdgpu_buffer_rsrc_t rsA[PARTS], rsB[PARTS];
#pragma unroll
    for (int p = 0; p < PARTS; ++p) {
        rsA[p] = tile_rsrc16(Xp + p * xplane + (size_t)m0 * ldx, (size_t)(M - m0) * ldx);
        rsB[p] = tile_rsrc16(Wp + p * wplane + (size_t)n0 * ldw, (size_t)(N - n0) * ldw);
    }
    unsigned voA[2], voB[NB];
#pragma unroll
    for (int i = 0; i < 2; ++i) voA[i] = (unsigned)((min(srow + RSTEP * i, M - 1 - m0) * ldx + sc8) * 2);
#pragma unroll
    for (int i = 0; i < NB; ++i) voB[i] = (unsigned)((min(srow + RSTEP * i, N - 1 - n0) * ldw + sc8) * 2);
    auto gload = [&](int kt, v4u (&ra)[PARTS][2], v4u (&rb)[PARTS][NB]) {
        const unsigned so = (unsigned)(k0 + kt * BK) * 2u;
#pragma unroll
        for (int p = 0; p < PARTS; ++p) {
            ra[p][0] = __builtin_amdgcn_raw_buffer_load_b128(rsA[p], voA[0], so, 0);
            ra[p][1] = __builtin_amdgcn_raw_buffer_load_b128(rsA[p], voA[1], so, 0);
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[p][i] = __builtin_amdgcn_raw_buffer_load_b128(rsB[p], voB[i], so, AZ_W_AUX);
        }
    };
    auto lstore = [&](int kt, int buf, const v4u (&ra)[PARTS][2], const v4u (&rb)[PARTS][NB]) {
        const bool ok = (k0 + kt * BK + sc8) < kend;     // K tail (K not a multiple of the chunk): zero
        unsigned short *st = lds + buf * STAGE;
#pragma unroll
        for (int p = 0; p < PARTS; ++p) {
            v4u a[2], b[NB];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = ra[p][i];
#pragma unroll
            for (int i = 0; i < NB; ++i) b[i] = rb[p][i];
            if (!ok) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = v4u{0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < NB; ++i) b[i] = v4u{0, 0, 0, 0};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
                *reinterpret_cast<v4u *>(st + p * ATILE + (srow + RSTEP * i) * LDR + sc8) = a[i];
#pragma unroll
            for (int i = 0; i < NB; ++i)
                *reinterpret_cast<v4u *>(st + PARTS * ATILE + p * BTILE + (srow + RSTEP * i) * LDR + sc8) = b[i];
        }
    };
    // fragments of one 16-wide k block h: A[r][p] (this wave's strips), B[p]
    auto frag = [&](int buf, int h, bf16x8 (&af)[NRTW > 0 ? NRTW : 1][PARTS], bf16x8 (&bf)[PARTS]) {
        const unsigned short *st = lds + buf * STAGE;
#pragma unroll
        for (int p = 0; p < PARTS; ++p) {
            bf[p] = *reinterpret_cast<const bf16x8 *>(st + PARTS * ATILE + p * BTILE + (cstrip * 32 + lrow) * LDR + 16 * h + lk);
#pragma unroll
            for (int r = 0; r < NRTW; ++r)
                af[r][p] = *reinterpret_cast<const bf16x8 *>(st + p * ATILE + ((wstrip0 + r) * 32 + lrow) * LDR + 16 * h + lk);
        }
    };
    // all cross terms of total order < PARTS, smallest first, x0*w0 last
    auto mfma16 = [&](const bf16x8 (&af)[NRTW > 0 ? NRTW : 1][PARTS], const bf16x8 (&bf)[PARTS]) {
#pragma unroll
        for (int ord = PARTS - 1; ord >= 0; --ord)
#pragma unroll
            for (int i = 0; i <= ord; ++i) {
                const int j = ord - i;
#pragma unroll
                for (int r = 0; r < NRTW; ++r) {
                    if constexpr (F16)
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[r][i]),
                                                                        __builtin_bit_cast(f16x8, bf[j]), acc[r], 0, 0, 0);
                    else
                        acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[r][i], bf[j], acc[r], 0, 0, 0);
                }
            }
    };

    v4u ra0[PARTS][2], rb0[PARTS][NB], ra1[PARTS][2], rb1[PARTS][NB];
    bf16x8 a0[NRTW > 0 ? NRTW : 1][PARTS], b0[PARTS], a1[NRTW > 0 ? NRTW : 1][PARTS], b1[PARTS];
    gload(0, ra0, rb0);
    gload(nk > 1 ? 1 : 0, ra1, rb1);
    __syncthreads();                 // the previous work item's readers are done with both buffers
    lstore(0, 0, ra0, rb0);
    __syncthreads();
    frag(0, 0, a0, b0);
    // Per step: [tile kt+1 -> LDS[buf^1] | request tile kt+2 | fragments of k block 1 | MFMAs of k block 0], barrier,
    // [fragments of k block 0 of tile kt+1 | MFMAs of k block 1].  The MFMAs lead each half and the other
    // instructions are dealt into their issue slots (sched_group_barrier), so the matrix pipe is fed from the first
    // cycle after the barrier (the fragments it needs were read before it) up to the last one before it.
    // (LDS[buf^1] is free for the store from the start of the step: its last readers -- k block 1 of the previous
    //  step's first half -- issued their reads before the previous step's barrier.)
    constexpr int NMF = NRTW * (PARTS * (PARTS + 1) / 2);            // MFMAs per k block
    constexpr int NRD = PARTS * (NRTW + 1);                          // fragment reads per k block
    constexpr int NWR = PARTS * (2 + NB);                            // tile vectors per thread (stores = loads)
    auto step = [&](int kt, int buf, v4u (&rl_a)[PARTS][2], v4u (&rl_b)[PARTS][NB], const v4u (&rw_a)[PARTS][2],
                    const v4u (&rw_b)[PARTS][NB]) {
        __builtin_amdgcn_sched_barrier(0);
        lstore(kt + 1, buf ^ 1, rw_a, rw_b);                // (past the last step: a tile nobody reads)
        gload(kt + 2 < nk ? kt + 2 : nk - 1, rl_a, rl_b);   // unconditional: exact vmcnt bookkeeping
        frag(buf, 1, a1, b1);
        mfma16(a0, b0);
        if constexpr (NMF > 0) {
            constexpr int per = (NWR + NWR + NRD + NMF - 1) / NMF;
#pragma unroll
            for (int i = 0; i < NMF; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // 1 MFMA, then up to `per` of:
                __builtin_amdgcn_sched_group_barrier(0x200 | 0x020 | 0x100, per, 0);   // DS write | VMEM read | DS read
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        frag(buf ^ 1, 0, a0, b0);
        mfma16(a1, b1);
        if constexpr (NMF > 0) {
            constexpr int per = (NRD + NMF - 1) / NMF;
#pragma unroll
            for (int i = 0; i < NMF; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, per, 1);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int kt = 0; kt < nk; kt += 2) {
        step(kt, 0, ra0, rb0, ra1, rb1);
        if (kt + 1 < nk) step(kt + 1, 1, ra1, rb1, ra0, rb0);
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5).
    const int col = n0 + cstrip * 32 + (lane & 31);
    if (col < N) {
#pragma unroll
        for (int r = 0; r < NRTW; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + (wstrip0 + r) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < M) slab[(size_t)row * N + col] = F16 ? acc[r][e] * oscale : acc[r][e];   // (a power of two: exact)
            }
    }
}

// Rows of one launch are cut into ceil(strips / 8) m-tiles of near-equal size.  A workgroup owns
// whole (n-tile, k-chunk) groups and walks their m-tiles; with one workgroup per CU the groups
// alive at any time cover half of the weight planes (205 MB for int6 with two terms), which the
// 256 MB Infinity Cache holds, so re-reading a weight panel for the next m-tile does not go to HBM.
template <int PARTS, int WAVES, bool F16>
__global__ void __launch_bounds__(WAVES * 64, 2)
k_fc_bf16(const unsigned short *__restrict__ Xp, int ldx, size_t xplane, const unsigned short *__restrict__ Wp,
          int ldw, size_t wplane, const int *Mptr, int capM, int N, int K, int S, int Kc,
          float *__restrict__ part, int min_strips, int max_strips, const float *__restrict__ scales, int xcd_order)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int M = *Mptr;
    if (M <= 0) return;
    const int strips = (M + 31) >> 5;
    if (strips < min_strips || strips > max_strips) return;    // the other shape of this kernel owns the launch
    constexpr int TS = WAVES;                     // strips per m-tile: 4 (128 rows) or 8 (256 rows)
    const int mt = (strips + TS - 1) / TS;
    const int nt = (N + BN - 1) / BN;
    const int G = nt * S;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float oscale = F16 ? scales[1] : 1.f;
    auto run = [&](int g, int mtile) {
        const int ntile = g / S, s = g - ntile * S;
        const int n0 = ntile * BN;
        const int k0 = s * Kc;
        const int kend = min(K, k0 + Kc);
        float *slab = part + (size_t)s * capM * N;
        int strip0, n_rt;
        mtile_rows(strips, mt, mtile, strip0, n_rt);
        const int m0 = strip0 * 32;
        // 8 waves: the live strips are split evenly between the two half-workgroups
        const int h0 = (WAVES == 8) ? (n_rt + 1) >> 1 : n_rt;
        const int wstrip0 = (wave >> 2) ? h0 : 0;
        const int nrtw = (wave >> 2) ? n_rt - h0 : h0;      // strips of this wave (0..4)
        switch (nrtw) {
        case 0: fc_tile_bf16<0, PARTS, WAVES, F16>(Xp, ldx, xplane, Wp, ldw, wplane, M, N, m0, wstrip0, n0, k0, kend, slab, lds, oscale); break;
        case 1: fc_tile_bf16<1, PARTS, WAVES, F16>(Xp, ldx, xplane, Wp, ldw, wplane, M, N, m0, wstrip0, n0, k0, kend, slab, lds, oscale); break;
        case 2: fc_tile_bf16<2, PARTS, WAVES, F16>(Xp, ldx, xplane, Wp, ldw, wplane, M, N, m0, wstrip0, n0, k0, kend, slab, lds, oscale); break;
        case 3: fc_tile_bf16<3, PARTS, WAVES, F16>(Xp, ldx, xplane, Wp, ldw, wplane, M, N, m0, wstrip0, n0, k0, kend, slab, lds, oscale); break;
        default: fc_tile_bf16<4, PARTS, WAVES, F16>(Xp, ldx, xplane, Wp, ldw, wplane, M, N, m0, wstrip0, n0, k0, kend, slab, lds, oscale); break;
        }
    };
    if (WAVES == 8 && xcd_order && mt > 1 && gridDim.x == 256 && (S & 7) == 0) {
        // (work items dealt per XCD so that the m-tiles of one weight panel run side by side: see k_fc_bf16x3)
        const int x = blockIdx.x & 7, q = blockIdx.x >> 3, per = gridDim.x >> 3;
        const int Gx = nt * (S >> 3);
        const bool rot = (per % mt) == 0;
        for (int i = q; i < Gx * mt; i += per) {
            const int j = i / mt;
            int t = i - j * mt;
            if (rot) t = (t + i / per) % mt;
            run((j % nt) * S + x + 8 * (j / nt), t);
        }
        return;
    }
    for (int g = blockIdx.x; g < G; g += gridDim.x)
        for (int mtile = 0; mtile < mt; ++mtile) run(g, mtile);
}


// ---------------------------------------------------------------------------------------------------------------
// Three terms per operand, six MFMAs per product (az_set_gemm_mode 3): fp32-grade arithmetic on the bf16 matrix
// cores.  x = x0 + x1 + x2 holds 24 mantissa bits exactly (8 per term), so the six cross terms of order <= 2
// leave out only x1*w2, x2*w1, x2*w2 (<= 2^-24 of the product each); accumulation is the MFMA's fp32.
//
// Same work decomposition as the two-term kernel above (K chunks -> slabs -> k_fc_reduce), other budget:
//   * LDS rows are 64 B (32 bf16) WITHOUT padding -- two stages of three A planes (256 rows) and three B planes
//     (128 rows) are 144 KB of the CU's 160 --, 16-byte vectors swizzled (vector v of row r sits at v ^ ((r >> 2) & 3)):
//     conflict-free for the fragments' ds_read_b128 (lane groups of 16 rows {0-3, 12-15, 20-27} / {4-11, 16-19,
//     28-31} each meet the 16 vector slots of a 256-byte bank row once) and for the 128-byte-contiguous stores;
//   * fragments are read per 16-wide k block, right before their MFMAs: with two waves per SIMD one wave's fragment
//     reads run under the other's 24 MFMAs, and the 60 fragment registers are not doubled;
//   * one barrier per K step: the stores of tile kt+1 go to the other stage at the top of step kt.
// Shapes: WAVES = 8 / AROWS = 256 (wave w: column strip w & 3, the row strips of half w >> 2), one workgroup per CU;
//         WAVES = 4 / AROWS = 64 for launches of <= 2 row strips (wave w: column strip w, every row strip), two per CU.
template <int NRTW, int WAVES, int AROWS>
__device__ __forceinline__ void fc_tile_x3(const unsigned short *__restrict__ Xp, int ldx, size_t xplane,
                                           const unsigned short *__restrict__ Wp, int ldw, size_t wplane, int M,
                                           int N, int m0, int wstrip0, int n0, int k0, int kend,
                                           float *__restrict__ slab, unsigned short *lds)
{
    constexpr int P = 3;
    constexpr int RS = WAVES * 16;                            // rows covered by one pass of all threads
    constexpr int NA = AROWS / RS, NB = BN / RS;              // vectors per thread per plane
    constexpr int ATILE = AROWS * BK, BTILE = BN * BK;        // one plane of A / of B (elements)
    constexpr int STAGE = P * (ATILE + BTILE);
    constexpr int NR = NRTW > 0 ? NRTW : 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cstrip = wave & 3;
    const int lrow = lane & 31;
    const int nk = (kend - k0 + BK - 1) / BK;

    floatx16 acc[NR];
#pragma unroll
    for (int r = 0; r < NRTW; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

    const int srow = tid >> 2, sc8 = (tid & 3) * 8;
    const int swz_st = (((tid & 3) ^ ((srow >> 2) & 3)) * 8);            // (RS is a multiple of 16: the same for every i)
    __amdgpu_buffer_rsrc_t rsA[P], rsB[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        rsA[p] = tile_rsrc16(Xp + p * xplane + (size_t)m0 * ldx, (size_t)(M - m0) * ldx);
        rsB[p] = tile_rsrc16(Wp + p * wplane + (size_t)n0 * ldw, (size_t)(N - n0) * ldw);
    }
    unsigned voA[NA], voB[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) voA[i] = (unsigned)((min(srow + RS * i, M - 1 - m0) * ldx + sc8) * 2);
#pragma unroll
    for (int i = 0; i < NB; ++i) voB[i] = (unsigned)((min(srow + RS * i, N - 1 - n0) * ldw + sc8) * 2);
    auto gload = [&](int kt, v4u (&ra)[P][NA], v4u (&rb)[P][NB]) {
        const unsigned so = (unsigned)(k0 + kt * BK) * 2u;
#pragma unroll
        for (int p = 0; p < P; ++p) {
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[p][i] = __builtin_amdgcn_raw_buffer_load_b128(rsA[p], voA[i], so, 0);
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[p][i] = __builtin_amdgcn_raw_buffer_load_b128(rsB[p], voB[i], so, AZ_W_AUX);
        }
    };
    auto lstore = [&](int kt, int buf, const v4u (&ra)[P][NA], const v4u (&rb)[P][NB]) {
        const bool ok = (k0 + kt * BK + sc8) < kend;     // K tail (K not a multiple of the chunk): zero
        unsigned short *st = lds + buf * STAGE;
#pragma unroll
        for (int p = 0; p < P; ++p) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                *reinterpret_cast<v4u *>(st + p * ATILE + (srow + RS * i) * BK + swz_st) = ok ? ra[p][i] : v4u{0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < NB; ++i)
                *reinterpret_cast<v4u *>(st + P * ATILE + p * BTILE + (srow + RS * i) * BK + swz_st) =
                    ok ? rb[p][i] : v4u{0, 0, 0, 0};
        }
    };
    const int fsw = (lrow >> 2) & 3, fhi = lane >> 5;
    auto kblock = [&](int buf, int h) {
        const unsigned short *st = lds + buf * STAGE;
        const int vo = (((2 * h + fhi) ^ fsw) * 8);
        bf16x8 bf[P], af[NR][P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            bf[p] = *reinterpret_cast<const bf16x8 *>(st + P * ATILE + p * BTILE + (cstrip * 32 + lrow) * BK + vo);
#pragma unroll
            for (int r = 0; r < NRTW; ++r)
                af[r][p] = *reinterpret_cast<const bf16x8 *>(st + p * ATILE + ((wstrip0 + r) * 32 + lrow) * BK + vo);
        }
        // cross terms of total order <= 2, smallest first, x0*w0 last
#pragma unroll
        for (int ord = P - 1; ord >= 0; --ord)
#pragma unroll
            for (int i = 0; i <= ord; ++i)
#pragma unroll
                for (int r = 0; r < NRTW; ++r)
                    acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[r][i], bf[ord - i], acc[r], 0, 0, 0);
    };

    v4u ra0[P][NA], rb0[P][NB], ra1[P][NA], rb1[P][NB];
    gload(0, ra0, rb0);
    gload(nk > 1 ? 1 : 0, ra1, rb1);
    __syncthreads();                 // the previous work item's readers are done with both stages
    lstore(0, 0, ra0, rb0);
    __syncthreads();
    auto step = [&](int kt, int buf, v4u (&rl_a)[P][NA], v4u (&rl_b)[P][NB], const v4u (&rw_a)[P][NA],
                    const v4u (&rw_b)[P][NB]) {
        lstore(kt + 1, buf ^ 1, rw_a, rw_b);                // (past the last step: a tile nobody reads)
        gload(kt + 2 < nk ? kt + 2 : nk - 1, rl_a, rl_b);   // unconditional: exact vmcnt bookkeeping
        kblock(buf, 0);
        kblock(buf, 1);
        __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) {
        step(kt, 0, ra0, rb0, ra1, rb1);
        if (kt + 1 < nk) step(kt + 1, 1, ra1, rb1, ra0, rb0);
    }

    const int col = n0 + cstrip * 32 + (lane & 31);
    if (col < N) {
#pragma unroll
        for (int r = 0; r < NRTW; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + (wstrip0 + r) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < M) slab[(size_t)row * N + col] = acc[r][e];
            }
    }
}

template <int WAVES, int AROWS>
__global__ void __launch_bounds__(WAVES * 64, 2)
k_fc_bf16x3(const unsigned short *__restrict__ Xp, int ldx, size_t xplane, const unsigned short *__restrict__ Wp,
            int ldw, size_t wplane, const int *Mptr, int capM, int N, int K, int S, int Kc,
            float *__restrict__ part, int min_strips, int max_strips, int xcd_order)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int M = *Mptr;
    if (M <= 0) return;
    const int strips = (M + 31) >> 5;
    if (strips < min_strips || strips > max_strips) return;    // the other shape of this kernel owns the launch
    constexpr int TS = AROWS / 32;                // strips per m-tile
    const int mt = (strips + TS - 1) / TS;
    const int nt = (N + BN - 1) / BN;
    const int G = nt * S;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto run = [&](int g, int mtile) {
        const int ntile = g / S, s = g - ntile * S;
        const int n0 = ntile * BN;
        const int k0 = s * Kc;
        const int kend = min(K, k0 + Kc);
        float *slab = part + (size_t)s * capM * N;
        int strip0, n_rt;
        mtile_rows(strips, mt, mtile, strip0, n_rt);
        const int m0 = strip0 * 32;
        const int h0 = (WAVES == 8) ? (n_rt + 1) >> 1 : n_rt;
        const int wstrip0 = (WAVES == 8 && (wave >> 2)) ? h0 : 0;
        const int nrtw = (WAVES == 8 && (wave >> 2)) ? n_rt - h0 : h0;      // strips of this wave
        switch (nrtw) {
        case 0: fc_tile_x3<0, WAVES, AROWS>(Xp, ldx, xplane, Wp, ldw, wplane, M, N, m0, wstrip0, n0, k0, kend, slab, lds); break;
        case 1: fc_tile_x3<1, WAVES, AROWS>(Xp, ldx, xplane, Wp, ldw, wplane, M, N, m0, wstrip0, n0, k0, kend, slab, lds); break;
        case 2: fc_tile_x3<2, WAVES, AROWS>(Xp, ldx, xplane, Wp, ldw, wplane, M, N, m0, wstrip0, n0, k0, kend, slab, lds); break;
        case 3: if constexpr (AROWS > 64) fc_tile_x3<3, WAVES, AROWS>(Xp, ldx, xplane, Wp, ldw, wplane, M, N, m0, wstrip0, n0, k0, kend, slab, lds); break;
        default: if constexpr (AROWS > 64) fc_tile_x3<4, WAVES, AROWS>(Xp, ldx, xplane, Wp, ldw, wplane, M, N, m0, wstrip0, n0, k0, kend, slab, lds); break;
        }
    };
    if (xcd_order && mt > 1 && gridDim.x == 256 && (S & 7) == 0) {
        // What limits this kernel is the operand feed from beyond the L2 (an XCD's 32 workgroups stream 32 different
        // weight panels).  Work items (group, m-tile) are dealt so that, on one XCD (workgroups b = x mod 8), the m-tiles
        // of a group run at the same time on neighbouring workgroups (one fetch of the weight panel serves mt readers
        // through the XCD's L2) and the groups of one round share a K chunk (so do their A panels).
        const int x = blockIdx.x & 7, q = blockIdx.x >> 3, per = gridDim.x >> 3;
        const int Gx = nt * (S >> 3);
        const bool rot = (per % mt) == 0;
        for (int i = q; i < Gx * mt; i += per) {
            const int j = i / mt;
            int t = i - j * mt;
            if (rot) t = (t + i / per) % mt;              // every workgroup meets every m-tile size
            const int s = x + 8 * (j / nt), ntile = j % nt;
            run(ntile * S + s, t);
        }
        return;
    }
    for (int g = blockIdx.x; g < G; g += gridDim.x)
        for (int mtile = 0; mtile < mt; ++mtile) run(g, mtile);
}

}  // namespace

// --------------------------------------------------------------------------------------
void azk_split_planes(hipStream_t s, const float *in, unsigned short *out, long long n, long long plane_stride,
                      int parts, float scale)
{
    hipLaunchKernelGGL(k_split_planes, dim3(4096), dim3(256), 0, s, in, out, n, plane_stride, parts, scale);
}

void azk_feat_scale(hipStream_t s, const float *feat, long long n, float *scales, float sw)
{
    hipLaunchKernelGGL(k_feat_scale, dim3(256), dim3(256), 0, s, feat, n, scales, sw);
}

// part[s][m][n] = sum over chunk s of X . W^T from two bf16 planes per operand.  Two shapes of one
// kernel, same per-row arithmetic (so a row's bits do not depend on which one ran):
//   <= 2 row strips: 128-row tiles, 4 waves, two workgroups per CU (weight-streaming bound: bytes
//                    in flight matter);
//   >= 3 row strips: 256-row tiles, 8 waves, one workgroup per CU (operand traffic per FLOP
//                    matters; weight panels are re-read from the Infinity Cache).
// Both are launched; each reads the row count on the device and one of them returns at once.
// Kc (elements) is the fp32 kernel's chunking, so the slabs feed the same k_fc_reduce.
int azk_fc_gemm_bf16(hipStream_t s, const unsigned short *Xp, int ldx, size_t xplane, const unsigned short *Wp,
                     int ldw, size_t wplane, const int *Mptr, int capM, int N, int K, int S, int Kc, float *part,
                     int parts, const float *scales)
{
    static const int xcd_order = getenv("AZ_X3_ORDER") ? atoi(getenv("AZ_X3_ORDER")) : 1;      // experiment knob
    if (parts == 3) {
        // three terms: 64-byte LDS rows (no padding), see fc_tile_x3
        const size_t shm_w = (size_t)2 * 3 * (256 + BN) * BK * sizeof(unsigned short);     // 144 KB
        const size_t shm_n = (size_t)2 * 3 * (64 + BN) * BK * sizeof(unsigned short);      // 72 KB
        static bool attr3 = false;
        if (!attr3) {
            if (hipFuncSetAttribute((const void *)k_fc_bf16x3<8, 256>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)shm_w) != hipSuccess) return -1;
            if (hipFuncSetAttribute((const void *)k_fc_bf16x3<4, 64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)shm_n) != hipSuccess) return -1;
            attr3 = true;
        }
        hipLaunchKernelGGL((k_fc_bf16x3<4, 64>), dim3(512), dim3(256), shm_n, s, Xp, ldx, xplane, Wp, ldw, wplane,
                           Mptr, capM, N, K, S, Kc, part, 1, 2, 0);
        hipLaunchKernelGGL((k_fc_bf16x3<8, 256>), dim3(256), dim3(512), shm_w, s, Xp, ldx, xplane, Wp, ldw, wplane,
                           Mptr, capM, N, K, S, Kc, part, 3, 1 << 30, xcd_order);
        return 0;
    }
    constexpr int PARTS = 2;
    const size_t shm_wide = (size_t)2 * PARTS * (256 + BN) * LDR * sizeof(unsigned short);
    const size_t shm_narrow = (size_t)2 * PARTS * (128 + BN) * LDR * sizeof(unsigned short);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)k_fc_bf16<PARTS, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm_wide) != hipSuccess) return -1;
#ifndef AZ_NO_NARROW
        if (hipFuncSetAttribute((const void *)k_fc_bf16<PARTS, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm_narrow) != hipSuccess) return -1;
#endif
        attr_done = true;
    }
    static const bool skip_narrow = getenv("AZ_BF16_SKIP_NARROW") != nullptr;   // experiment knob
#ifndef AZ_NO_NARROW
    if (!skip_narrow)
    hipLaunchKernelGGL((k_fc_bf16<PARTS, 4, true>), dim3(512), dim3(256), shm_narrow, s, Xp, ldx, xplane, Wp, ldw, wplane,
                       Mptr, capM, N, K, S, Kc, part, 1, 2, scales, 0);
#endif
    hipLaunchKernelGGL((k_fc_bf16<PARTS, 8, true>), dim3(256), dim3(512), shm_wide, s, Xp, ldx, xplane, Wp, ldw, wplane,
                       Mptr, capM, N, K, S, Kc, part, 3, 1 << 30, scales, xcd_order);
    return 0;
}
