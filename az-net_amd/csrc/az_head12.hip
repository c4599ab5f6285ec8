// InnerProduct as split-K GEMM on the fp32 matrix cores, many-row shape: one weight tile feeds up to 12 row strips.
//
// k_fc_splitk (az_head.hip) gives a workgroup of 4 waves a 128-row m-tile, two workgroups per CU; a group (n-tile,
// K-chunk) walks its m-tiles one after the other and re-reads its weight panel for each: 6 times at the 688 rows of
// a one-pass search (az_static.hip) = 2.5 GB of weight traffic per launch against 0.41 GB of weights.
// Here a workgroup is 12 waves = 3 row groups x 4 column strips, ONE workgroup per CU (3 waves per SIMD -- one more
// than the 2 x 4-wave arrangement, so the matrix pipe has more cover -- and the CU's whole LDS for one tile pair):
// the m-tile is up to 384 rows (12 strips, <= 4 per row group), the 128 x 32 weight tile of a K-step is staged
// once for all of them, and 688 rows are two m-tiles of 11 strips: the weights are read twice, not six times.
// Arithmetic is k_fc_splitk's, instruction for instruction per output element: v_mfma_f32_32x32x2_f32 over the same
// fixed K chunks in the same k order (0,4,1,5,2,6,3,7 inside each 8-wide group), one slab per chunk -- a row's bits
// do not depend on which kernel (or which m-tile) computed it (tests/test_gpu_parity.py).
// Per K-step and wave: 4 k-groups x (<= 4 strips x 4) MFMAs; fragments of the next k-group are read while the current
// group's MFMAs run (two fragment sets); tiles are requested two K-steps ahead with ONE register set (freed when its
// tile goes to LDS during the third k-group, re-used for the request right behind the step's one barrier, which sits
// before the fourth k-group, whose operands are already in registers); work items are chained through LDS (the
// last steps of an item fetch the next item's first tile).
// Inside a k-group the LDS / VMEM instructions are dealt one per MFMA issue slot with the MFMAs in front
// (sched_group_barrier): 1012 -> 985 us.  The launch's last <= 16 rows are a 16x16x4 half strip (az_head.hip's, same
// bits) in the row group that ends the last m-tile: 982 -> 972 us (without the pacing it cost 10 us instead).
// Measured at 688 rows (config A, one pass): 972 us against 1022-1035 for k_fc_splitk on the same boxes,
// FETCH_SIZE x 2 + WRITE_SIZE = 1.13 GB per launch against 2.79 GB; no LDS bank conflicts.
// Tried here and dropped: s_setprio around the MFMA groups, either polarity (+60 us).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "az_dev.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int W_NT = 768;                 // 12 waves
constexpr int W_BM = 384, W_BN = 128, W_BK = 32;
constexpr int W_LDT = W_BK + 4;           // padded LDS row (floats): 144 B, conflict-free b128 reads
constexpr int W_NLA = W_BM * (W_BK / 4) / W_NT;      // 4 activation float4s per thread per K-step
constexpr int W_NLB = 2;                  // weight float4s per thread per K-step (1024 over 768 threads: guarded)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const float *base, size_t elems_left)
{
    const size_t bytes = elems_left * sizeof(float);
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes,
                                             0x00020000);
}

__device__ __forceinline__ float4 ld128(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

__device__ __forceinline__ float4 zero_if(float4 v, bool ok)
{
    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
    return v;
}

// One (m-tile, n-tile, K-chunk) work item.
struct Item12 {
    int m0, nst, n0, k0, kend, t;       // t: which m-tile of the launch
    float *slab;
};

// NRT = strips of THIS wave's row group (the three row groups of a tile differ by at most one strip; waves take
// different instantiations, every one with the same staging work and the same barriers).
// Work items are chained: the last K-step of an item requests the FIRST tile of the workgroup's next item (instead
// of a tile nobody reads) and stages it in the LDS buffer that step leaves free, so the next item starts with its
// operands already in LDS[pb] (`preloaded`) -- with one workgroup per CU nothing else would cover that latency.
// Returns the buffer holding the next item's first tile.
// HALF: the row group ends with a 16-row half strip (the launch's last <= 16 rows) on v_mfma_f32_16x16x4_f32, fed so
// that its accumulation chain is the 32x32x2 one (az_head.hip: per 8-wide k group two instructions, k slots
// (0,4,1,5) then (2,6,3,7)): half the matrix-pipe time of a padded strip, same bits.
template <int NRT, bool HALF>
__device__ __forceinline__ int tile12(const float *__restrict__ X, int ldx, const float *__restrict__ Wt, int ldw,
                                      int M, int N, const Item12 &it, int my0, const Item12 &nx, bool has_next,
                                      bool preloaded, int pb, float *sA, float *sB)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cs = wave & 3;
    const int lrow = lane & 31, lk = (lane >> 5) * 4;
    const int nk = (it.kend - it.k0 + W_BK - 1) / W_BK;
    static_assert(!HALF || NRT <= 3, "a row group has at most four strip slots");
    constexpr int NA = NRT > 0 ? NRT : 1;
    floatx4 acch[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    floatx16 acc[NA];
#pragma unroll
    for (int r = 0; r < NRT; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

    // global -> register staging: thread t owns float4 f = t + 768 i: row f / 8, columns (f % 8) * 4 .. +3
    // (rows past the tile's strips are staged too -- clamped loads, LDS rows nobody reads: no per-thread predicates)
    const int srow = tid >> 3, sc4 = (tid & 7) * 4;
    const int KT = (ldw + W_BK - 1) / W_BK;                  // tile-major weights (k_tile_weights)
    // (activation rows past M are outside the descriptor: they read as 0 and land in output rows that are never
    //  stored -- no clamp, so the per-thread offsets are the same for every item)
    struct Src { __amdgpu_buffer_rsrc_t rsA, rsB; int k0; };
    auto src_of = [&](const Item12 &q) {
        Src d;
        d.rsA = rsrc_of(X + (size_t)q.m0 * ldx, (size_t)(M - q.m0) * ldx);
        d.rsB = rsrc_of(Wt + (size_t)(q.n0 / W_BN) * KT * (W_BN * W_BK), (size_t)KT * (W_BN * W_BK));
        d.k0 = q.k0;
        return d;
    };
    const Src cur = src_of(it), nxt = src_of(has_next ? nx : it);
    unsigned voA[W_NLA];
#pragma unroll
    for (int i = 0; i < W_NLA; ++i) voA[i] = (unsigned)(((srow + 96 * i) * ldx + sc4) * 4);
    unsigned voB[W_NLB];
#pragma unroll
    for (int i = 0; i < W_NLB; ++i) voB[i] = (unsigned)(((min(srow + 96 * i, W_BN - 1)) * W_BK + sc4) * 4);
    const bool liveB1 = wave < 4;                            // second weight vector: f = t + 768 < 1024 (wave-uniform)
    auto gload = [&](const Src &d, int kt, float4 (&ra)[W_NLA], float4 (&rb)[W_NLB]) {
        const unsigned so = (unsigned)(d.k0 + kt * W_BK) * 4u;
        const unsigned sob = (unsigned)(d.k0 / W_BK + kt) * (unsigned)(W_BN * W_BK * 4);
#pragma unroll
        for (int i = 0; i < W_NLA; ++i) ra[i] = ld128(d.rsA, voA[i], so);
#pragma unroll
        for (int i = 0; i < W_NLB; ++i) rb[i] = ld128(d.rsB, voB[i], sob);
    };
    // (K and the chunk length are multiples of the K-step here -- the launcher checks --, so no tail masking.
    //  The second weight vector exists for waves 0-3 only; the others store theirs to a junk slot: a conditional
    //  store would let the compiler sink the LOAD into the branch, right in front of its use.)
    float *junk = sB + 2 * W_BN * W_LDT + (tid & 511) * 4;
    auto lstore = [&](int buf, const float4 (&ra)[W_NLA], const float4 (&rb)[W_NLB]) {
        float *a = sA + buf * (W_BM * W_LDT), *b = sB + buf * (W_BN * W_LDT);
#pragma unroll
        for (int i = 0; i < W_NLA; ++i) *reinterpret_cast<float4 *>(&a[(srow + 96 * i) * W_LDT + sc4]) = ra[i];
        *reinterpret_cast<float4 *>(&b[srow * W_LDT + sc4]) = rb[0];
        *reinterpret_cast<float4 *>(liveB1 ? &b[(srow + 96) * W_LDT + sc4] : junk) = rb[1];
    };
    const float *a_base = sA + (my0 * 32 + lrow) * W_LDT + lk;
    const float *b_base = sB + (cs * 32 + lrow) * W_LDT + lk;
    auto frag = [&](int buf, int g8, float4 (&af)[NA], float4 &bf) {
        if constexpr (NRT > 0) {
            bf = *reinterpret_cast<const float4 *>(b_base + buf * (W_BN * W_LDT) + g8 * 8);
#pragma unroll
            for (int r = 0; r < NRT; ++r)
                af[r] = *reinterpret_cast<const float4 *>(a_base + buf * (W_BM * W_LDT) + r * 32 * W_LDT + g8 * 8);
        }
    };
    // 4 * NRT MFMAs on one 8-wide k group; k-pairs {j, 4 + j}: accumulation order 0,4,1,5,2,6,3,7
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

    // Tiles are staged two K-steps ahead with ONE register set: the set is free once its tile is in LDS (k group
    // 2), so the request for the tile after next goes out right behind the barrier (k group 3) and has four k groups
    // of three waves (~12 000 cycles) to come back.  Across work items only the LDS tile is carried: the step before
    // last requests the next item's first tile, the last step stages it (and requests nothing, so that no load is in
    // flight through the epilogue), and an item that starts with its tile 0 in LDS[pb] requests tile 1 at once.
    // half-strip fragments of one 8-wide k group: lane = (row or column lane & 15, k slot lane >> 4);
    // slot s feeds k = {0,4,1,5}[s] to the first instruction and k + 2 to the second
    struct HalfFrag { float a1, a2, b1[2], b2[2]; };
    const int hk = ((lane >> 4) & 1) * 4 + (lane >> 5);
    const float *ha_base = sA + ((my0 + NRT) * 32 + (lane & 15)) * W_LDT + hk;
    const float *hb_base = sB + (cs * 32 + (lane & 15)) * W_LDT + hk;
    auto hfrag = [&](int buf, int g8, HalfFrag &f) {
        if constexpr (HALF) {
            const float *pa = ha_base + buf * (W_BM * W_LDT) + g8 * 8;
            f.a1 = pa[0]; f.a2 = pa[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float *pbh = hb_base + buf * (W_BN * W_LDT) + h * 16 * W_LDT + g8 * 8;
                f.b1[h] = pbh[0]; f.b2[h] = pbh[2];
            }
        }
    };
    auto hmfma = [&](const HalfFrag &f) {
        if constexpr (HALF) {
#pragma unroll
            for (int h = 0; h < 2; ++h) acch[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a1, f.b1[h], acch[h], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h) acch[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a2, f.b2[h], acch[h], 0, 0, 0);
        }
    };

    float4 ra[W_NLA], rb[W_NLB];
    if (!preloaded) {                // (workgroup-uniform) the workgroup's first item
        gload(cur, 0, ra, rb);
        lstore(pb, ra, rb);
        __syncthreads();
    }
    gload(cur, 1, ra, rb);
    float4 a0[NA], a1[NA], b0, b1;
    HalfFrag h0, h1;
    frag(pb, 0, a0, b0);
    hfrag(pb, 0, h0);
    // one K-step on LDS[buf]; the tile in flight is staged into LDS[buf^1], then tile `gkt` of `g` is requested
    // Inside a k group the other instructions are dealt one per MFMA issue slot (sched_group_barrier), the MFMAs
    // in front: the matrix pipe is fed from the first cycle after the barrier and up to the last one before it.
    // (a half strip adds 4 short MFMAs and 3 two-word DS reads per group)
    constexpr int NM = 4 * NRT + (HALF ? 4 : 0), NR = NRT + 1 + (HALF ? 3 : 0), NV = W_NLA + W_NLB;
    constexpr bool PACE = NM >= NV + NR;
    constexpr bool PACE2 = !PACE && NM > 0;     // few MFMAs (1-2 strips): several other instructions per MFMA slot
    constexpr int NMD = NM > 0 ? NM : 1;
    auto step = [&](int buf, const Src &g, int gkt, bool request, bool prefetch_frag) {
        // k group 0 | fragments of group 1
        __builtin_amdgcn_sched_barrier(0);
        frag(buf, 1, a1, b1);
        hfrag(buf, 1, h1);
        mfma8(a0, b0);
        hmfma(h0);
        if constexpr (PACE) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
        } else if constexpr (PACE2) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, (NR + NMD - 1) / NMD, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // k group 1 | fragments of group 2
        frag(buf, 2, a0, b0);
        hfrag(buf, 2, h0);
        mfma8(a1, b1);
        hmfma(h1);
        if constexpr (PACE) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 1);
        } else if constexpr (PACE2) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, (NR + NMD - 1) / NMD, 1);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // k group 2 | staged tile -> LDS[buf^1] | fragments of group 3
        lstore(buf ^ 1, ra, rb);
        frag(buf, 3, a1, b1);
        hfrag(buf, 3, h1);
        mfma8(a0, b0);
        hmfma(h0);
        if constexpr (PACE) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 2);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 2);   // 1 DS write
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 2);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 2);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NM - NV - NR, 2);
        } else if constexpr (PACE2) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 2);
                __builtin_amdgcn_sched_group_barrier(0x200 | 0x100, (NV + NR + NMD - 1) / NMD, 2);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();             // LDS[buf^1] complete; everyone's reads of LDS[buf] issued
        __builtin_amdgcn_sched_barrier(0);
        // k group 3 | request the tile after next | fragments of group 0 of the next tile
        if (request) gload(g, gkt, ra, rb);
        if (prefetch_frag) { frag(buf ^ 1, 0, a0, b0); hfrag(buf ^ 1, 0, h0); }
        mfma8(a1, b1);
        hmfma(h1);
        if constexpr (PACE) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 3);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 3);   // 1 VMEM read
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 3);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 3);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NM - NV - NR, 3);
        } else if constexpr (PACE2) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 3);
                __builtin_amdgcn_sched_group_barrier(0x020 | 0x100, (NV + NR + NMD - 1) / NMD, 3);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int kt = 0; kt + 2 < nk; ++kt) step((kt & 1) ^ pb, cur, kt + 2, true, true);
    step(((nk - 2) & 1) ^ pb, nxt, 0, true, true);     // (no next item: a tile nobody reads)
    step(((nk - 1) & 1) ^ pb, nxt, 0, false, false);

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8*(e >> 2) + 4*(lane >> 5).
    const int col = it.n0 + cs * 32 + (lane & 31);
    if (col < N) {
#pragma unroll
        for (int r = 0; r < NRT; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = it.m0 + (my0 + r) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < M) it.slab[(size_t)row * N + col] = acc[r][e];
            }
    }
    if constexpr (HALF) {
        // C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + e
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int hcol = it.n0 + cs * 32 + 16 * h + (lane & 15);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = it.m0 + (my0 + NRT) * 32 + 4 * (lane >> 4) + e;
                if (hcol < N && row < M) it.slab[(size_t)row * N + hcol] = acch[h][e];
            }
        }
    }
    return (nk & 1) ^ pb;
}

__global__ void __launch_bounds__(W_NT)
k_fc_splitk12(const float *__restrict__ X, int ldx, const float *__restrict__ Wt, int ldw, const int *Mptr, int capM,
              int N, int K, int S, int Kc, float *__restrict__ part, int min_rows, int pair_mode, unsigned long long *ts)
{
    extern __shared__ __attribute__((aligned(16))) float lds12[];
    float *sA = lds12, *sB = lds12 + 2 * W_BM * W_LDT;
    const int M = *Mptr;
    if (M <= 0 || M < min_rows) return;      // (fewer rows: k_fc_splitk, launched beside this kernel, owns the launch)
    AzSpan span(ts);                         // (profiling: first workgroup in, last workgroup out -- az_dev.h)
    // strip slots: full strips, then (<= 16 trailing rows) one half-strip slot, the last slot of the last m-tile
    const bool has_half = (M & 31) != 0 && (M & 31) <= 16;
    const int strips = (M + 31) >> 5;
    const int mt = (strips + 11) / 12;
    const int nt = (N + W_BN - 1) / W_BN;
    const int G = nt * S;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rg = wave >> 2;
    // the workgroup's items: its groups g = blockIdx.x, + gridDim.x, ...; inside a group the m-tiles
    const int ngrp = ((int)blockIdx.x < G) ? (G - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int nitems = ngrp * mt;
    const int tb = strips / mt, tr = strips - tb * mt;       // strips dealt evenly to the m-tiles
    // Two m-tiles (the 350 - 768 row launches): the two readers of a group's weight panel are workgroups b and b + 8
    // -- the same XCD, so the same L2 -- working on it AT THE SAME TIME (one takes m-tile 0, the other m-tile 1, roles
    // swapping from group to group so that the 11- and the 10-strip tile do not let one drift ahead): the second read of
    // a weight tile then finds it in L2 instead of going to memory again.  Same items, same bits; only who does which.
    // (every workgroup then has G / (grid / 2) items -- the ngrp * mt it has anyway when the groups divide evenly)
    const bool paired = pair_mode && mt == 2 && ((int)gridDim.x & 15) == 0 && G % (int)gridDim.x == 0;
    const int pq = ((int)blockIdx.x / 16) * 8 + ((int)blockIdx.x & 7), prole = ((int)blockIdx.x >> 3) & 1;
    auto item_at = [&](int idx) {
        int gi = idx / mt, t = idx - gi * mt;
        int g = (int)blockIdx.x + gi * (int)gridDim.x;
        if (paired) {
            // the pair's groups pq, pq + grid/2, ...: one item of each per workgroup, the m-tile alternating
            g = pq + idx * ((int)gridDim.x / 2);
            t = (idx + prole) & 1;
        }
        const int ntile = g / S, s = g - ntile * S;
        Item12 q;
        q.t = t;
        q.m0 = (t * tb + (t < tr ? t : tr)) * 32;
        q.nst = tb + (t < tr ? 1 : 0);
        q.n0 = ntile * W_BN; q.k0 = s * Kc; q.kend = min(K, q.k0 + Kc);
        q.slab = part + (size_t)s * capM * N;
        return q;
    };
    int pb = 0;
    for (int idx = 0; idx < nitems; ++idx) {
        const Item12 it = item_at(idx);
        const bool has_next = idx + 1 < nitems;
        const Item12 nx = item_at(has_next ? idx + 1 : idx);
        // ... then to the three row groups of a tile
        const int rb_ = it.nst / 3, rr = it.nst - rb_ * 3;
        const int my0 = rg * rb_ + (rg < rr ? rg : rr);
        int nrt = rb_ + (rg < rr ? 1 : 0);
        // the half slot: last slot of the launch's last m-tile, i.e. of the row group that ends at the tile's end
        const bool half = has_half && it.t == mt - 1 && nrt > 0 && my0 + nrt == it.nst;
        if (half) --nrt;
#define TILE12(NRT_, HALF_) pb = tile12<NRT_, HALF_>(X, ldx, Wt, ldw, M, N, it, my0, nx, has_next, idx > 0, pb, sA, sB)
        if (half) {
            switch (nrt) {
            case 0: TILE12(0, true); break;
            case 1: TILE12(1, true); break;
            case 2: TILE12(2, true); break;
            default: TILE12(3, true); break;
            }
        } else {
            switch (nrt) {
            case 0: TILE12(0, false); break;
            case 1: TILE12(1, false); break;
            case 2: TILE12(2, false); break;
            case 3: TILE12(3, false); break;
            default: TILE12(4, false); break;
            }
        }
#undef TILE12
    }
}

}  // namespace

// One workgroup per CU; the caller knows the row count on the host (a one-pass search's plan) and takes this
// kernel for many-row launches of layers whose (n-tile, K-chunk) groups fill the chip.
static size_t lds12_bytes() { return (size_t)(2 * W_BM * W_LDT + 2 * W_BN * W_LDT + 512 * 4) * sizeof(float); }   // + junk slots

// The kernel's 149 KB of dynamic LDS need an opt-in that is recorded per DEVICE: every context calls this once with
// its device current (az_load_head) and keeps k_fc_splitk for all launches if it fails.
int azk_fc_gemm12_prepare()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_fc_splitk12), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds12_bytes()) == hipSuccess ? 0 : -1;
}

void azk_fc_gemm12(hipStream_t s, const float *x, int ldx, const float *W, int ldw, const int *Mptr, int capM, int N,
                   int K, int S, int Kc, float *part, int min_rows, unsigned long long *ts)
{
    static int grid = -1, pair = -1;    // AZ_GEMM12_GRID, AZ_GEMM12_PAIR=0: environment switches, the same for every device
    if (grid < 0) { const char *e = getenv("AZ_GEMM12_GRID"); grid = e ? atoi(e) : 256; }
    if (pair < 0) { const char *e = getenv("AZ_GEMM12_PAIR"); pair = (e && !atoi(e)) ? 0 : 1; }
    hipLaunchKernelGGL(k_fc_splitk12, dim3(grid), dim3(W_NT), lds12_bytes(), s, x, ldx, W, ldw, Mptr, capM, N, K, S, Kc, part, min_rows,
                       pair, ts);
}
