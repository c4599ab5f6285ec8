// az_head_terms.hip -- the int6 GEMM on the 16-bit matrix cores with fp32 operands written as sums of 16-bit terms
// (az_set_gemm_mode 2 / 3; the default, mode 0, is the fp32-input MFMA of az_head.hip / az_head12.hip).
//
// gfx950's fp32-input MFMA runs at 1/16 of the fp16 / bf16 MFMA rate.  An fp32 value is a sum of 16-bit terms, each the
// rounding of what the previous ones left, and the product of two such sums, accumulated in fp32 by the MFMA, needs
//   mode 2:  two fp16 terms of x * 2^k (11 + 11 mantissa bits; 2^k brings the largest |weight| resp. the largest
//            |feature-map value| of the image to [2^14, 2^15): nothing overflows, and a power of two scales exactly):
//            x0*w1 + x1*w0 + x0*w0                                 3 MFMAs, products good to ~2^-21;
//   mode 3:  three bf16 terms (8 + 8 + 8 bits: every fp32 value exactly, fp32's exponent range):
//            x1*w1 + x0*w2 + x2*w0 + x0*w1 + x1*w0 + x0*w0        6 MFMAs, what is left out is <= 2^-23 of a product.
// Measured at the full head, outputs against an f64 evaluation (tests/test_gpu_gemm_modes.py, bench.py): both modes are
// as close as the fp32-MFMA path itself (~1e-6).  The terms ("planes") are produced once: for the weights when the
// head is loaded (k_split_planes), for pool5 by the RoIPool kernel; mode 2's per-image scale by k_feat_scale.
//
// Same fixed K chunks -> slabs -> k_fc_reduce as the fp32 kernels, and ONE arithmetic for every launch size (a roi's
// bits do not depend on its batch).  With the matrix pipe 5x / 2.7x cheaper, what binds is operand delivery (LDS
// reads / stores per MFMA, L2 -> CU traffic), hence the tile shapes below.
#include "az_dev.h"
#include <hip/hip_fp16.h>
#include <stdlib.h>
#include <type_traits>

#ifndef AZ_W_AUX
#define AZ_W_AUX 0          /* buffer-load cache policy of the weight stream (2 = nt measured 25 % slower) */
#endif

namespace {

constexpr int BN = 128, BK = 32;                // BK in 16-bit elements = two 32x32x16 MFMA k blocks

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

// Scale of the fp16 terms of pool5 (two-term mode): pool5 values are maxima of feature-map values, so
// |pool5| <= max |map|; sc[0] = sx = the power of two that brings that maximum into [2^14, 2^15), sc[1] = 1 / (sx * sw)
// (what the GEMM's epilogue multiplies its sums by; sw = the weights' power-of-two scale).  One launch: block maxima ->
// atomicMax on the bits (|x| >= 0: unsigned order = float order) -> the last block to arrive writes the scales and
// clears the scratch words sc[2], sc[3] for the next call.
__global__ void __launch_bounds__(1024) k_feat_scale(const float *__restrict__ feat, long long n, float *sc, float sw)
{
    __shared__ float wm[16];
    float m = 0.f;
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    // (four independent loads in flight per thread: the loop is a chain of memory round trips otherwise)
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = reinterpret_cast<const float4 *>(feat)[i], b = reinterpret_cast<const float4 *>(feat)[i + stride],
                     c = reinterpret_cast<const float4 *>(feat)[i + 2 * stride], d = reinterpret_cast<const float4 *>(feat)[i + 3 * stride];
        const float ma = fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w)));
        const float mb = fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)));
        const float mc = fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w)));
        const float md = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)));
        m = fmaxf(m, fmaxf(fmaxf(ma, mb), fmaxf(mc, md)));
    }
    for (; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(feat)[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    if (blockIdx.x == 0)
        for (long long t = (n4 << 2) + threadIdx.x; t < n; t += blockDim.x) m = fmaxf(m, fabsf(feat[t]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, wm[w]);
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

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ void mtile_rows(int strips, int mt, int t, int &strip0, int &nstrips)
{
    const int base = strips / mt, rem = strips - base * mt;
    strip0 = t * base + (t < rem ? t : rem);
    nstrips = base + (t < rem ? 1 : 0);
}

// ---------------------------------------------------------------------------------------------------------------
// One (m-tile, n-tile, K-chunk) work item on P planes per operand.
//   * LDS rows are 64 B (32 terms) WITHOUT padding -- two stages of three A planes (256 rows) and three B planes
//     (128 rows) are 144 KB of the CU's 160 --, 16-byte vectors swizzled (vector v of row r sits at v ^ ((r >> 2) & 3)):
//     conflict-free for the fragments' ds_read_b128 (lane groups of 16 rows {0-3, 12-15, 20-27} / {4-11, 16-19,
//     28-31} each meet the 16 vector slots of a 256-byte bank row once) and for the 128-byte-contiguous stores
//     (SQ_LDS_BANK_CONFLICT = 0, profiles/);
//   * strip-pipelined K step (see below): fragment reads, tile stores and global requests are dealt into the MFMA
//     issue slots; one barrier per K step;
//   * work items are dealt per XCD so that the m-tiles of one weight panel run side by side (k_fc_terms).
// Shapes: WAVES = 8 / AROWS = 256: wave w owns column strip(s) w & 3 and the row strips of half w >> 2; one workgroup
//           per CU.  CW = 1: 256 x 128 tiles (three terms: LDS);  CW = 2: 256 x 256 tiles, two column strips per
//           wave (two terms: a third fewer LDS bytes per MFMA, what that mode is bound by: 480 -> 410 us at 670 rows);
//         WAVES = 4 / AROWS = 64 for launches of <= 2 row strips (wave w: column strip w, every row strip), two per CU.
#ifndef AZ_TERMS_DMA
#define AZ_TERMS_DMA 1          /* 1: tiles go global -> LDS directly (buffer_load_dwordx4 ... lds), no staging registers, no
                                   ds_write pass; 0: through registers, requested two K steps ahead.  Measured at 670 rows:
                                   direct 364 / 629 us (two / three terms), through registers 378 / 655 */
#endif
template <int NRTW, int WAVES, int AROWS, int P, bool F16, int CW>
__device__ __forceinline__ void fc_tile_terms(const unsigned short *__restrict__ Xp, int ldx, size_t xplane,
                                              const unsigned short *__restrict__ Wp, int ldw, size_t wplane, int M,
                                              int N, int m0, int wstrip0, int n0, int k0, int kend,
                                              float *__restrict__ slab, unsigned short *lds, float oscale)
{
    constexpr int RS = WAVES * 16;                            // rows covered by one pass of all threads
    constexpr int BNT = BN * CW;                              // columns of the tile: 128, or 256 (two strips per wave)
    constexpr int NA = AROWS / RS, NB = BNT / RS;             // vectors per thread per plane
    constexpr int ATILE = AROWS * BK, BTILE = BNT * BK;       // one plane of A / of B (elements)
    constexpr int STAGE = P * (ATILE + BTILE);
    constexpr int NR = NRTW > 0 ? NRTW : 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cstrip = wave & 3;
    const int lrow = lane & 31;
    const int nk = (kend - k0 + BK - 1) / BK;
    const int KT = (ldw + BK - 1) / BK;                       // K steps of a whole weight row (ldw = K)
    const int KTA = (ldx + BK - 1) / BK;                      // ... of a whole activation row

    floatx16 acc[NR][CW];
#pragma unroll
    for (int r = 0; r < NRTW; ++r)
#pragma unroll
        for (int c = 0; c < CW; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;

    const int srow = tid >> 2, sc8 = (tid & 3) * 8;
    __amdgpu_buffer_rsrc_t rsA[P], rsB[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        // (tile-major activation planes: block (row / 32, k / 32) is 2 KB; rows past M inside the last strip hold stale
        //  finite terms that only reach output rows nobody stores, strips past it are outside the descriptor: zeros)
        rsA[p] = tile_rsrc16(Xp + p * xplane + (size_t)(m0 >> 5) * KTA * 1024, (size_t)(((M + 31) >> 5) - (m0 >> 5)) * KTA * 1024);
        // (tile-major weight planes: block (n0 / 128 + c, k / 32) is 8 KB; the descriptor spans this tile's CW block rows)
        rsB[p] = tile_rsrc16(Wp + p * wplane + (size_t)(n0 / BN) * KT * (BN * BK),
                             (size_t)min(CW, (N + BN - 1) / BN - n0 / BN) * KT * (BN * BK));      // (past the last block row: zeros)
    }
    unsigned voA[NA], voB[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int arow = srow + RS * i;
        voA[i] = (unsigned)(((size_t)(arow >> 5) * KTA * 1024 + (arow & 31) * 32 + sc8) * 2);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int brow = srow + RS * i;                       // row of the B tile: block row brow / 128, row brow % 128 inside
        voB[i] = (unsigned)(((size_t)(brow / BN) * KT * (BN * BK) + (brow % BN) * BK + sc8) * 2);
    }
    auto gload = [&](int kt, v4u (&ra)[P][NA], v4u (&rb)[P][NB]) {
        const unsigned soa = (unsigned)(k0 / BK + kt) * 2048u;
        const unsigned sob = (unsigned)(k0 / BK + kt) * (unsigned)(BN * BK * 2);      // (chunks start on K-step boundaries)
#pragma unroll
        for (int p = 0; p < P; ++p) {
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[p][i] = __builtin_amdgcn_raw_buffer_load_b128(rsA[p], voA[i], soa, 0);
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[p][i] = __builtin_amdgcn_raw_buffer_load_b128(rsB[p], voB[i], sob, AZ_W_AUX);
        }
    };
    // (the planes are tile-major, zero-padded in K and pre-swizzled: the stage is a linear image of the tile, and a K that
    //  is not a multiple of the K step needs no masking -- chunks start on K-step boundaries, the last one ends in zeros)
    auto lstore = [&](int kt, int buf, const v4u (&ra)[P][NA], const v4u (&rb)[P][NB]) {
        (void)kt;
        unsigned short *st = lds + buf * STAGE;
#pragma unroll
        for (int p = 0; p < P; ++p) {
#pragma unroll
            for (int i = 0; i < NA; ++i) *reinterpret_cast<v4u *>(st + p * ATILE + (srow + RS * i) * BK + sc8) = ra[p][i];
#pragma unroll
            for (int i = 0; i < NB; ++i) *reinterpret_cast<v4u *>(st + P * ATILE + p * BTILE + (srow + RS * i) * BK + sc8) = rb[p][i];
        }
    };
    // direct form: one wave instruction moves the 1 KB (16 rows x 64 B) its lanes would have staged, to the same place
    auto dma = [&](int kt, int buf) {
        typedef __attribute__((address_space(3))) void *lds_ptr;
        unsigned short *st = lds + buf * STAGE + wave * 16 * BK;
        const unsigned soa = (unsigned)(k0 / BK + kt) * 2048u;
        const unsigned sob = (unsigned)(k0 / BK + kt) * (unsigned)(BN * BK * 2);
#pragma unroll
        for (int p = 0; p < P; ++p) {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA[p], (lds_ptr)(st + p * ATILE + RS * i * BK), 16, voA[i], soa, 0, 0);
#pragma unroll
            for (int i = 0; i < NB; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB[p], (lds_ptr)(st + P * ATILE + p * BTILE + RS * i * BK), 16, voB[i], sob, 0,
                                                         AZ_W_AUX);
        }
    };
    const int fsw = (lrow >> 2) & 3, fhi = lane >> 5;

    // Strip-pipelined form: a K step is 2 * NRTW phases (k block h, row strip r), each six MFMAs on one accumulator
    // tile; the fragments of the NEXT phase are read while the current phase's MFMAs run (two A sets of three
    // planes, one B set per k block), so LDS reads are spread evenly under the matrix pipe instead of coming in a
    // block of 15 in front of 24 MFMAs.  The step's one barrier sits in front of the LAST phase, whose operands
    // are in registers by then: behind it that phase prefetches phase 0 of the next step from the other stage.
    // The tile stores (stage buf^1) and the global requests (tile kt + 2) are dealt into the first phases.
    // Tiles are requested two K steps ahead with two register sets (the set whose tile has just gone to LDS is re-used
    // at once for the tile two steps on; set 0: even tiles, set 1: odd tiles) -- or, where registers are short (the
    // 256-column shape), one step ahead with one set.
    constexpr bool DMA = (AZ_TERMS_DMA != 0) && P > 0;       // (value-dependent, so that the unused branch is discarded)
    constexpr bool TWO = CW == 1 && !DMA;
    v4u ra0[P][NA], rb0[P][NB], ra1[TWO ? P : 1][NA], rb1[TWO ? P : 1][NB];
    if constexpr (DMA) {
        __syncthreads();             // the previous work item's readers are done with both stages
        dma(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
        gload(0, ra0, rb0);
        if constexpr (TWO) gload(nk > 1 ? 1 : 0, ra1, rb1);
        __syncthreads();                 // the previous work item's readers are done with both stages
        lstore(0, 0, ra0, rb0);
        __syncthreads();
        if constexpr (TWO) gload(nk > 2 ? 2 : nk - 1, ra0, rb0);
        else gload(nk > 1 ? 1 : 0, ra0, rb0);
    }
    constexpr int NPH = 2 * NRTW;
    bf16x8 bq[2][CW][P], aq[2][P];
    auto rdA = [&](int buf, int h, int r, bf16x8 (&a)[P]) {
        const unsigned short *st = lds + buf * STAGE;
        const int vo = (((2 * h + fhi) ^ fsw) * 8);
#pragma unroll
        for (int p = 0; p < P; ++p)
            a[p] = *reinterpret_cast<const bf16x8 *>(st + p * ATILE + ((wstrip0 + r) * 32 + lrow) * BK + vo);
    };
    auto rdB = [&](int buf, int h, bf16x8 (&b)[CW][P]) {
        const unsigned short *st = lds + buf * STAGE;
        const int vo = (((2 * h + fhi) ^ fsw) * 8);
#pragma unroll
        for (int c = 0; c < CW; ++c)
#pragma unroll
            for (int p = 0; p < P; ++p)
                b[c][p] = *reinterpret_cast<const bf16x8 *>(st + P * ATILE + p * BTILE + ((cstrip * CW + c) * 32 + lrow) * BK + vo);
    };
    // the cross terms of total order < P, smallest first, x0 * w0 last
    constexpr int NMF = CW * P * (P + 1) / 2;
    auto mf6 = [&](int r, const bf16x8 (&a)[P], const bf16x8 (&b)[CW][P]) {
#pragma unroll
        for (int ord = P - 1; ord >= 0; --ord)
#pragma unroll
            for (int i = 0; i <= ord; ++i)
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    if constexpr (F16)
                        acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i]),
                                                                           __builtin_bit_cast(f16x8, b[c][ord - i]), acc[r][c], 0, 0, 0);
                    else
                        acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[c][ord - i], acc[r][c], 0, 0, 0);
                }
    };
    if constexpr (NRTW > 0) { rdB(0, 0, bq[0]); rdA(0, 0, 0, aq[0]); }
    constexpr int AHEAD = TWO ? 3 : 2;
    auto step = [&](int kt, int buf, auto &rs_a, auto &rs_b) {     // the set holding tile kt + 1
        if constexpr (NRTW == 0) {
            if constexpr (DMA) {
                dma(kt + 1 < nk ? kt + 1 : nk - 1, buf ^ 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                lstore(kt + 1, buf ^ 1, rs_a, rs_b);
                gload(kt + AHEAD < nk ? kt + AHEAD : nk - 1, rs_a, rs_b);
            }
            __syncthreads();
        } else {
            constexpr int NW = P * (NA + NB);                     // tile vectors per thread: stores = requests
            static_for<0, NPH>([&](auto phc) {
                constexpr int ph = decltype(phc)::value;
                constexpr int h = ph / NRTW, r = ph % NRTW;
                constexpr bool last = ph == NPH - 1;
                constexpr int nh = (ph + 1) / NRTW, nr = (ph + 1) % NRTW;
                constexpr bool st_here = ph == 0 && !DMA, ld_here = DMA ? ph == 0 : ph == (NPH > 1 ? 1 : 0);
                constexpr int nother = P + (st_here ? NW : 0) + (ld_here ? NW : 0) + ((last || nh != h) ? CW * P : 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (last) {
                    if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (st_here) lstore(kt + 1, buf ^ 1, rs_a, rs_b);     // (past the last step: a tile nobody reads)
                if constexpr (ld_here) {
                    if constexpr (DMA) dma(kt + 1 < nk ? kt + 1 : nk - 1, buf ^ 1);
                    else gload(kt + AHEAD < nk ? kt + AHEAD : nk - 1, rs_a, rs_b);   // unconditional: exact vmcnt bookkeeping
                }
                if constexpr (!last) {
                    rdA(buf, nh, nr, aq[(ph + 1) & 1]);
                    if constexpr (nh != h) rdB(buf, nh, bq[nh]);
                } else {
                    rdA(buf ^ 1, 0, 0, aq[0]);                     // NPH is even: phase 0 of the next step uses set 0
                    rdB(buf ^ 1, 0, bq[0]);
                }
                mf6(r, aq[ph & 1], bq[h]);
                constexpr int per = (nother + NMF - 1) / NMF;
#pragma unroll
                for (int i = 0; i < NMF; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100 | 0x200 | 0x020, per, 0);
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int kt = 0; kt < nk; kt += 2) {
        if constexpr (TWO) {
            step(kt, 0, ra1, rb1);
            if (kt + 1 < nk) step(kt + 1, 1, ra0, rb0);
        } else {
            step(kt, 0, ra0, rb0);
            if (kt + 1 < nk) step(kt + 1, 1, ra0, rb0);
        }
    }
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        const int col = n0 + (cstrip * CW + c) * 32 + (lane & 31);
        if (col < N) {
#pragma unroll
            for (int r = 0; r < NRTW; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = m0 + (wstrip0 + r) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (row < M) slab[(size_t)row * N + col] = F16 ? acc[r][c][e] * oscale : acc[r][c][e];   // (a power of two: exact)
                }
        }
    }
}

template <int WAVES, int AROWS, int P, bool F16, int CW>
__global__ void __launch_bounds__(WAVES * 64, 2)
k_fc_terms(const unsigned short *__restrict__ Xp, int ldx, size_t xplane, const unsigned short *__restrict__ Wp,
           int ldw, size_t wplane, const int *Mptr, int capM, int N, int K, int S, int Kc,
           float *__restrict__ part, int min_strips, int max_strips, int xcd_order, const float *__restrict__ scales)
{
    extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
    const int M = *Mptr;
    if (M <= 0) return;
    const int strips = (M + 31) >> 5;
    if (strips < min_strips || strips > max_strips) return;    // the other shape of this kernel owns the launch
    constexpr int TS = AROWS / 32;                // strips per m-tile
    const int mt = (strips + TS - 1) / TS;
    constexpr int BNT = BN * CW;
    const int nt = (N + BNT - 1) / BNT;
    const int G = nt * S;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float oscale = F16 ? scales[1] : 1.f;
    auto run = [&](int g, int mtile) {
        const int ntile = g / S, s = g - ntile * S;
        const int n0 = ntile * BNT;
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
        case 0: fc_tile_terms<0, WAVES, AROWS, P, F16, CW>(Xp, ldx, xplane, Wp, ldw, wplane, M, N, m0, wstrip0, n0, k0, kend, slab, lds, oscale); break;
        case 1: fc_tile_terms<1, WAVES, AROWS, P, F16, CW>(Xp, ldx, xplane, Wp, ldw, wplane, M, N, m0, wstrip0, n0, k0, kend, slab, lds, oscale); break;
        case 2: fc_tile_terms<2, WAVES, AROWS, P, F16, CW>(Xp, ldx, xplane, Wp, ldw, wplane, M, N, m0, wstrip0, n0, k0, kend, slab, lds, oscale); break;
        case 3: if constexpr (AROWS > 64) fc_tile_terms<3, WAVES, AROWS, P, F16, CW>(Xp, ldx, xplane, Wp, ldw, wplane, M, N, m0, wstrip0, n0, k0, kend, slab, lds, oscale); break;
        default: if constexpr (AROWS > 64) fc_tile_terms<4, WAVES, AROWS, P, F16, CW>(Xp, ldx, xplane, Wp, ldw, wplane, M, N, m0, wstrip0, n0, k0, kend, slab, lds, oscale); break;
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
// Weight planes for k_fc_terms: tile-major -- block (n / 128, k / 32) holds 128 rows x 32 terms (8 KB) contiguously, K padded
// with zeros to a multiple of 32, N to a multiple of 128 -- so that a K step's weight tile is one contiguous 8 KB
// read per plane instead of 128 pieces of 64 B that are a weight row (tens of KB) apart.
size_t azk_weight_plane_elems(int N, int K)
{
    return (size_t)((N + BN - 1) / BN) * BN * (size_t)((K + BK - 1) / BK) * BK;
}

namespace {
__global__ void k_split_planes_tiled(const float *__restrict__ in, unsigned short *__restrict__ out, int N, int K,
                                     long long plane_stride, int parts, float scale)
{
    const int KT = (K + BK - 1) / BK;
    const long long total = (long long)((N + BN - 1) / BN) * BN * KT * BK;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        // (o = position in the plane; inside a row the four 16-byte vectors are XOR-swizzled the way the LDS tile is
        //  read, so global -> LDS is a linear copy: position vector pv of row r holds k vector pv ^ ((r >> 2) & 3))
        const int pk = (int)(o % BK), r = (int)((o / BK) % BN);
        const long long blk = o / (BK * BN);
        const int kt = (int)(blk % KT), nt = (int)(blk / KT);
        const int kk = ((((pk >> 3) ^ ((r >> 2) & 3)) << 3) | (pk & 7));
        const int n = nt * BN + r, k = kt * BK + kk;
        float x = (n < N && k < K) ? in[(size_t)n * K + k] : 0.f;
        if (scale != 0.f) {
            x *= scale;
            for (int p = 0; p < parts; ++p) {
                const __half h = __float2half_rn(x);
                out[p * plane_stride + o] = __half_as_ushort(h);
                x -= __half2float(h);
            }
        } else {
            for (int p = 0; p < parts; ++p) {
                const unsigned short h = f2bf(x);
                out[p * plane_stride + o] = h;
                x -= bf2f(h);
            }
        }
    }
}
}  // namespace

void azk_split_weight_planes(hipStream_t s, const float *in, unsigned short *out, int N, int K, int parts, float scale)
{
    hipLaunchKernelGGL(k_split_planes_tiled, dim3(4096), dim3(256), 0, s, in, out, N, K,
                       (long long)azk_weight_plane_elems(N, K), parts, scale);
}

void azk_feat_scale(hipStream_t s, const float *feat, long long n, float *scales, float sw)
{
    hipLaunchKernelGGL(k_feat_scale, dim3(128), dim3(1024), 0, s, feat, n, scales, sw);
}

// part[s][m][n] = sum over chunk s of X . W^T from the operands' planes of 16-bit terms.  Two shapes of one kernel, same
// per-row arithmetic (a row's bits do not depend on which one ran): <= 2 row strips: 64-row tiles, 4 waves, two
// workgroups per CU (weight streaming); >= 3 row strips: 256-row tiles, 8 waves, one workgroup per CU.  Both are launched;
// each reads the row count on the device and one of them returns at once.  Kc (elements) is the fp32 kernels' chunking,
// so the slabs feed the same k_fc_reduce.
namespace {
template <int P, bool F16, int CW> struct TermsShape {
    static constexpr size_t shm_w = (size_t)2 * P * (256 + BN * CW) * BK * sizeof(unsigned short);   // 144 KB (P = 3), 128 KB (P = 2)
    static constexpr size_t shm_n = (size_t)2 * P * (64 + BN) * BK * sizeof(unsigned short);         // 72 KB, 48 KB
    static int prepare()
    {
        if (hipFuncSetAttribute((const void *)k_fc_terms<8, 256, P, F16, CW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm_w) != hipSuccess) return -1;
        if (hipFuncSetAttribute((const void *)k_fc_terms<4, 64, P, F16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shm_n) != hipSuccess) return -1;
        return 0;
    }
    static void launch(hipStream_t s, const unsigned short *Xp, int ldx, size_t xplane, const unsigned short *Wp, int ldw,
                       size_t wplane, const int *Mptr, int capM, int N, int K, int S, int Kc, float *part, int xcd_order,
                       const float *scales)
    {
        hipLaunchKernelGGL((k_fc_terms<4, 64, P, F16, 1>), dim3(512), dim3(256), shm_n, s, Xp, ldx, xplane, Wp, ldw, wplane,
                           Mptr, capM, N, K, S, Kc, part, 1, 2, 0, scales);
        hipLaunchKernelGGL((k_fc_terms<8, 256, P, F16, CW>), dim3(256), dim3(512), shm_w, s, Xp, ldx, xplane, Wp, ldw, wplane,
                           Mptr, capM, N, K, S, Kc, part, 3, 1 << 30, xcd_order, scales);
    }
};
typedef TermsShape<2, true, 2> Terms2;       // two fp16 terms: 256 x 256 tiles
typedef TermsShape<3, false, 1> Terms3;      // three bf16 terms: 256 x 128 tiles
}  // namespace

// The kernels' dynamic LDS (> 64 KB) needs an opt-in that is recorded per DEVICE: every context calls this once with its
// device current (az_load_head); != 0: the mode cannot run there.
int azk_fc_terms_prepare(int parts)
{
    return parts == 2 ? Terms2::prepare() : parts == 3 ? Terms3::prepare() : -1;
}

int azk_fc_gemm_terms(hipStream_t s, const unsigned short *Xp, int ldx, size_t xplane, const unsigned short *Wp,
                      int ldw, size_t wplane, const int *Mptr, int capM, int N, int K, int S, int Kc, float *part,
                      int parts, const float *scales)
{
    static const int xcd_order = getenv("AZ_X3_ORDER") ? atoi(getenv("AZ_X3_ORDER")) : 1;      // experiment knob
    if (parts == 3) Terms3::launch(s, Xp, ldx, xplane, Wp, ldw, wplane, Mptr, capM, N, K, S, Kc, part, xcd_order, scales);
    else if (parts == 2) Terms2::launch(s, Xp, ldx, xplane, Wp, ldw, wplane, Mptr, capM, N, K, S, Kc, part, xcd_order, scales);
    else return -1;
    return 0;
}
