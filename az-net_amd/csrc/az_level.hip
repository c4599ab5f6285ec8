// az_level.hip -- everything between the head evaluations of two consecutive levels of the search loop, as
// ONE single-workgroup kernel (lib/detect/test.py:373-391 with 189-257 and lib/utils/div.pyx:15-89 inlined):
//
//   candidates of level l   _unwrap_adj_pred: MIN_SIDE filter, ordered append to Y / aScores
//   zoom selection          zoom[0] = 1 at level 1; indZ = where(zoom >= Tz); Z = B[indZ]
//   divide_region(Z)        children, then _sift_dup = np.unique over the 10-px hash (first occurrence,
//                           ascending hash order) -> B of level l+1
//   level l+1's rois        _get_rois_blob + the feature-space dedup np.unique(return_index, return_inverse)
//
// The multi-launch form of the same steps (az_geom.hip: ten launches per level, each a wave per element
// over the whole chip) stays as the path for levels that outgrow this kernel's LDS tables, for the last level
// (whose candidate copy and top-k are data-heavy: one CU moves ~25 GB/s) and as the bit-for-bit cross-check in
// the tests.  A level in the middle of the tree is a few hundred regions and a few KB of state: every stage
// boundary of the multi-launch form costs a kernel boundary plus a first-touch round trip to memory another
// XCD has just written (~4-5 us), ten times per level.  Here the level's inputs (inv_index, zoom scores, keep
// flags) come in with ONE round trip, everything else runs out of LDS with a barrier between stages, and both
// np.unique calls are an LDS bucket sort of (hash << bits | position) words: run heads are the unique values
// in ascending order and, positions being the low bits, the head of a run is its first occurrence -- the same
// integers as the O(N^2) rank kernels.  Same device helpers (az_geom_dev.h) for the f64 arithmetic, built with
// -ffp-contract=off.
#include "az_geom_dev.h"
#include <stdlib.h>

namespace {

#ifndef AZ_LV_NT
#define AZ_LV_NT 1024
#endif
constexpr int NT = AZ_LV_NT;
constexpr int LV_R = 1024;         // regions per level handled here
constexpr int LV_C = 4096;         // children per level (before _sift_dup)

// LDS carve-up in 8-byte words (one buffer; stages that share a region never overlap in time):
//   [0, 4096) sort words   [4096, 6144) sczi   [6144, 6656) szr   [6656, 7168) zoom scores   [7168, 8576) keep flags
//     ([4096, 8192): dead once the next level's regions exist -- the sort scratch of the pair-speculation stage)
//   [8576, 9088) schoff (divide) / all-children offsets (pair speculation)
//   [9088, 9600) inv, then the provenance (all-children index) of the next level's regions
//   [9600, 13696) sort scratch, then the next level's regions (f64 x 4 x 1024)
//   [13696, 15745) sort buckets
constexpr int W_SORT = 0, W_SCZI = 4096, W_SZR = 6144, W_ZOOM = 6656, W_KEEP = 7168, W_CHOFF = 8576, W_INV = 9088,
              W_BN = 9600, W_BINS = 13696;
constexpr int W_TMP2 = 4096;
constexpr int LDS_WORDS = W_BINS + (SORT_NB + 2) / 2 + 1;
static_assert(W_KEEP + (LV_R * AZ_NSUB + 7) / 8 <= W_CHOFF, "keep flags overlap the next region");
static_assert(W_TMP2 + LV_C <= W_CHOFF, "pair-speculation sort scratch overlaps live data");

#ifdef AZ_LEVEL_TIMING
#define TSTAMP() do { __syncthreads(); if (tid == 0 && tsn < 32) ts[tsn++] = wall_clock64(); } while (0)
#define TREPORT() do { if (tid == 0) { printf("level %d stamps (x10ns):", l); for (int i = 1; i < tsn; ++i) printf(" %llu", ts[i] - ts[i - 1]); printf("\n"); } } while (0)
#else
#define TSTAMP() do { } while (0)
#define TREPORT() do { } while (0)
#endif

static __device__ __forceinline__ void level_geom_body(const AzLevelArgs &a)
{
#ifdef AZ_LEVEL_TIMING
    __shared__ unsigned long long ts[32];
    int tsn = 0;
#endif
    __shared__ __attribute__((aligned(16))) unsigned long long sbuf[LDS_WORDS];
    __shared__ int wsum[17];
    __shared__ unsigned s_mm[2];
    const int tid = threadIdx.x;
    AzCounts *cnt = a.cnt;
    const int l = a.level;
    // Two workgroups: workgroup 0 walks the critical chain (zoom selection, divide_region, _sift_dup, the next level's
    // rois, ...) and writes every counter; workgroup 1 copies this level's candidates to Y / aScores, which nothing
    // before the final selection reads -- ~10 us of dependent global round trips off the path to the next head pass.
    // Both read the same inputs.  The chain writes the NEXT level's index / inv_index / rois / counters; the only one of
    // those that shares a name with an input is inv_index, which is why a.inv (this level's) and a.inv_next are two
    // buffers: the copier may be dispatched late (CUs held by another context's GEMM) and read a.inv at any time.
    const bool copier = gridDim.x == 1 || blockIdx.x == 1, chain = blockIdx.x == 0;
    const int P = cnt->P[l];
    const int U = *a.Uptr;
    const int ybase = cnt->ytot[l];
    if (cnt->err & (8 | 2048)) return;                     // an earlier fused stage overflowed (2048: a batch's pass did): the host reruns
    // (an overflow also records the level: the host then keeps the levels before it on the fused kernels)
    if (P > LV_R || U + a.root_row > LV_R) { if (tid == 0 && chain) { atomicOr(&cnt->err, 8); cnt->scratch[5] = l + 1; } return; }
    // the deferred root is the LAST row of this level's head pass (behind any pair-speculation rows)
    const int root_u = a.root_row ? cnt->PR[l] - 1 : 0;
    const int spec_base = cnt->SPB[l];                     // (lookup_next) first pair-speculation row of this level's pass
    TSTAMP();

    // ---- one round trip: this level's inv_index, zoom scores and keep flags -> LDS; its regions -> cache ----
    int *sinv = reinterpret_cast<int *>(sbuf + W_INV);
    float *szoom = reinterpret_cast<float *>(sbuf + W_ZOOM);
    unsigned char *skeep = reinterpret_cast<unsigned char *>(sbuf + W_KEEP);
    const double *B = a.B;
    {
        double warm = 0.0;
        for (int r = tid; r < P; r += NT) { sinv[r] = a.inv[r]; warm += B[4 * (size_t)r]; }
        for (int u = tid; u < U; u += NT) szoom[u] = a.zoom_u[u];
        for (int i = tid; i < U * AZ_NSUB; i += NT) skeep[i] = a.keep_u[i];
        if (a.root_row && tid < AZ_NSUB) skeep[U * AZ_NSUB + tid] = a.keep_u[(size_t)root_u * AZ_NSUB + tid];
        if (warm == -1.2345e300) szoom[0] = 0.f;           // (never true; keeps the region loads alive: they warm this CU's caches)
    }
    __syncthreads();
    TSTAMP();

    // ---- candidates (test.py:171-187, 380-381): pred_u holds the boxes decoded against the unique roi's own
    //      anchor and keep_u their MIN_SIDE filter (tail kernel); region r reads the row of its unique roi
    //      inv[r].  Thread t owns the contiguous candidates [t*per, (t+1)*per): ONE block scan orders them. ----
    int nc;
    {
        const int NC = P * AZ_NSUB;
        const int per = (NC + NT - 1) / NT;                 // <= 11 (P <= LV_R)
        const int c0 = tid * per, c1 = min(NC, c0 + per);
        unsigned keep = 0;
        for (int c = c0; c < c1; ++c) {
            const int r = c / AZ_NSUB, sub = c - r * AZ_NSUB;
            if (skeep[sinv[r] * AZ_NSUB + sub]) keep |= 1u << (c - c0);
        }
        int tot;
        int dst = ybase + block_excl_scan(__popc(keep), &tot, wsum);
        // the copy, in two batches of independent loads (the boxes of 12 candidates do not fit the registers
        // of a 1024-thread workgroup)
#pragma unroll
        for (int h = 0; h < 2 && 6 * h < per && copier; ++h) {
            double bx[6][4];
            float sc[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int c = min(c0 + 6 * h + j, NC - 1);
                const int r = c / AZ_NSUB, sub = c - r * AZ_NSUB;
                const size_t src = (size_t)sinv[r] * AZ_NSUB + sub;
                const double *pb = a.pred_u + src * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) bx[j][q] = pb[q];
                sc[j] = a.score_u[src];
            }
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (keep & (1u << (6 * h + j))) {
                    if (dst < a.capCand) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) a.Yall[(size_t)dst * 4 + q] = bx[j][q];
                        a.Sall[dst] = sc[j];
                    }
                    ++dst;
                }
        }
        nc = tot;
    }
    if (ybase + nc > a.capCand) { nc = a.capCand - ybase; if (tid == 0 && chain) atomicOr(&cnt->err, 2); }
    int shift = 0;
    if (a.root_row && !copier) {
        // (the chain workgroup only needs the number of root candidates the MIN_SIDE filter kept)
        const int kp = (tid < AZ_NSUB) ? (int)skeep[U * AZ_NSUB + tid] : 0;
        const unsigned long long m = __ballot(kp);
        if (tid == 0) wsum[0] = __popcll(m);
        __syncthreads();
        shift = AZ_NSUB - wsum[0];
        __syncthreads();
        if (tid == 0) cnt->NC[0] = AZ_NSUB - shift;
    } else if (a.root_row) {
        // The root's row rode on this level's head pass (az_fused.hip: its candidates are the first of Y, and the
        // 11 slots at the head of Y / aScores were reserved for them): the kept ones go there in order, and if
        // the MIN_SIDE filter dropped some, everything behind closes the gap.
        __syncthreads();                                    // (this level's candidates are in place)
        const int kp = (tid < AZ_NSUB) ? (int)skeep[U * AZ_NSUB + tid] : 0;
        const unsigned long long m = __ballot(kp);          // (wave 0 holds the 11 flags)
        if (tid == 0) wsum[0] = __popcll(m);
        if (tid < AZ_NSUB && kp) {
            const int dst = __popcll(m & ((1ull << tid) - 1ull));
            const size_t src = (size_t)root_u * AZ_NSUB + tid;
#pragma unroll
            for (int q = 0; q < 4; ++q) a.Yall[(size_t)dst * 4 + q] = a.pred_u[src * 4 + q];
            a.Sall[dst] = a.score_u[src];
        }
        __syncthreads();
        const int nc0 = wsum[0];
        shift = AZ_NSUB - nc0;
        __syncthreads();
        if (shift > 0) {
            const int end = ybase + nc;
            for (int base = AZ_NSUB; base < end; base += NT) {
                const int i = base + tid;
                double bx[4] = {0.0, 0.0, 0.0, 0.0};
                float sc = 0.f;
                if (i < end) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) bx[q] = a.Yall[(size_t)i * 4 + q];
                    sc = a.Sall[i];
                }
                __syncthreads();
                if (i < end) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) a.Yall[(size_t)(i - shift) * 4 + q] = bx[q];
                    a.Sall[i - shift] = sc;
                }
                __syncthreads();
            }
        }
        // (ytot[1 .. l] keep the layout with the 11 reserved slots: nothing reads them again; ytot[l + 1] below is
        //  the closed-up count)
        if (tid == 0 && chain) cnt->NC[0] = nc0;
    }
    if (!chain) return;
    TSTAMP();

    // ---- zoom selection (test.py:383-387) -----------------------------------------------------------
    int *szr = reinterpret_cast<int *>(sbuf + W_SZR), *schoff = reinterpret_cast<int *>(sbuf + W_CHOFF);
    int *sczi_w = reinterpret_cast<int *>(sbuf + W_SCZI);
    int PZ = 0;
    for (int base = 0; base < P; base += NT) {
        const int r = base + tid;
        int zf = 0;
        if (r < P) {
            float z = szoom[sinv[r]];
            if (a.force_root && l == 0 && r == 0) z = 1.0f;
            zf = ((double)z >= a.Tz);
        }
        int tot;
        const int off = block_excl_scan(zf, &tot, wsum);
        if (zf) szr[PZ + off] = r;
        PZ += tot;
    }
    if (tid == 0) { cnt->NC[l] = nc; cnt->ytot[l + 1] = ybase + nc - shift; cnt->PZ[l] = PZ; }
    __syncthreads();

    // ---- divide_region (div.pyx:15-76) -----------------------------------------------------------
    int CH = 0;
    for (int base = 0; base < PZ; base += NT) {
        const int z = base + tid;
        const int n = z < PZ ? div_nchildren(div_plan(B + 4 * (size_t)szr[z])) : 0;
        int tot;
        const int ex = block_excl_scan(n, &tot, wsum);
        if (z < PZ) {
            schoff[z] = CH + ex;
            if (CH + ex + n <= LV_C)
                for (int bi = 0; bi < n; ++bi) sczi_w[CH + ex + bi] = (z << 16) | bi;        // (child -> parent, child number)
        }
        CH += tot;
    }
    if (CH > LV_C || CH > a.capCh) { if (tid == 0) { atomicOr(&cnt->err, 8); cnt->scratch[5] = l + 1; } return; }
    unsigned long long *ssort = sbuf + W_SORT, *stmp = sbuf + W_BN;
    unsigned *sbins = reinterpret_cast<unsigned *>(sbuf + W_BINS);
    int *sczi = reinterpret_cast<int *>(sbuf + W_SCZI);
    __syncthreads();
    // (one thread per CHILD: a parent's children one after the other are ~35 dependent f64 divisions)
    for (int ci = tid; ci < CH; ci += NT) {
        const int z = sczi[ci] >> 16, bi = sczi[ci] & 0xFFFF;
        const double *r = B + 4 * (size_t)szr[z];
        double c[4];
        const long long key = div_child(r, div_plan(r), bi, a.min_side, c);
        ssort[ci] = ((unsigned long long)key << 20) | (unsigned)ci;                      // key < 1000^4 < 2^40
    }
    TSTAMP();
    // ---- _sift_dup (div.pyx:78-89) ------------------------------------------------------------------
    block_bucket_sort(ssort, CH, stmp, sbins, 40, wsum, s_mm);              // high part = hash >> 20
    TSTAMP();
    double *sBn = reinterpret_cast<double *>(sbuf + W_BN);                  // (the sort is done with its scratch)
    int *sprov = sinv;                                                      // (inv_index of this level is no longer needed)
    int Pn = 0;
    for (int base = 0; base < CH; base += NT) {
        const int i = base + tid;
        int head = 0;
        unsigned long long w = 0;
        if (i < CH) {
            w = ssort[i];
            head = (i == 0) || ((ssort[i - 1] >> 20) != (w >> 20));
        }
        int tot;
        const int ex = block_excl_scan(head, &tot, wsum);
        if (head) {
            const int slot = Pn + ex;
            if (slot < LV_R && slot < a.capR) {
                const int ci = sczi[(int)(w & 0xFFFFFu)];
                const double *r = B + 4 * (size_t)szr[ci >> 16];
                double c[4];
                div_child(r, div_plan(r), ci & 0xFFFF, a.min_side, c);
#pragma unroll
                for (int q = 0; q < 4; ++q) { sBn[4 * slot + q] = c[q]; a.Bnext[4 * (size_t)slot + q] = c[q]; }
                // which child of which region of THIS level it is, as an index into the all-children list the previous
                // geometry kernel laid out: finds the region's row among this level's pair-speculation rows
                if (a.lookup_next == 1) sprov[slot] = a.choff_all[szr[ci >> 16]] + (ci & 0xFFFF);
            }
        }
        Pn += tot;
    }
    if (Pn > LV_R || Pn > a.capR) { if (tid == 0) { atomicOr(&cnt->err, Pn > a.capR ? 1 : 8); cnt->scratch[5] = l + 1; } return; }
    if (tid == 0) { cnt->CH[l] = CH; cnt->P[l + 1] = Pn; }
    if (a.cut_next) {                // the host stopped the search after this level: right if the tree did, too
        if (tid == 0 && Pn > 0) atomicOr(&cnt->err, 1024);
        return;
    }
    __syncthreads();
    TSTAMP();
    // ---- level l+1: roi projection + feature-space dedup (test.py:61-97, 210-218) ---------------------
    if (Pn > a.batch) { if (tid == 0) { atomicOr(&cnt->err, 8); cnt->scratch[5] = l + 1; } return; }      // chunked dedup: multi-launch path
    int *sidx = reinterpret_cast<int *>(sbuf + W_SZR);                      // (szr is done) index[] of level l+1, in LDS
    const int Un = roi_dedup_sorted(sBn, Pn, a.scale, a.dedup, ssort, ssort + LV_R, sbins, s_mm, wsum, nullptr, a.index,
                                    a.inv_next, a.urois, a.ubox, sidx);
    if (tid == 0) cnt->U[l + 1] = Un;
    __syncthreads();
    TSTAMP();
    if (a.lookup_next == 2) {
        // ---- whole-tree speculation: level l+1's head outputs by RoIPool window among the rows of the search's one head
        //      pass (the full tree of this image shape; table built once per shape, az_static.hip).  Same arithmetic as
        //      the pair-row lookup below; a window the table does not hold sends the search back to the other form.
        // (one probe per unique roi, its row parked in LDS; then one thread per (roi, sub-region) decodes)
        int miss = 0;
        for (int slot = tid; slot < Un; slot += NT) {
            const int rpos = sidx[slot];
            float roi5[5];
            roi5[0] = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) roi5[1 + q] = (float)(sBn[4 * rpos + q] * a.scale);
            const int row = az_tab_lookup(a.stab, a.stabT, roi5, a.spatial_scale, a.root_row_full);
            sprov[slot] = row;
            if (row < 0) miss = 1; else a.zoom_v[slot] = a.zoom_all[row];
        }
        __syncthreads();
        for (int i = tid; i < Un * AZ_NSUB; i += NT) {
            const int slot = i / AZ_NSUB, sub = i - slot * AZ_NSUB;
            const int rpos = sidx[slot];
            const int row = sprov[slot];
            if (row < 0) continue;
            float d4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) d4[q] = a.delta_u[(size_t)row * 4 * AZ_NSUB + 4 * sub + q];
            double bx[4];
            az_decode_box(sBn + 4 * rpos, d4, a.im_h, a.im_w, a.eps, bx);
#pragma unroll
            for (int q = 0; q < 4; ++q) a.pred_v[(size_t)i * 4 + q] = bx[q];
            const float sc = a.score_all[(size_t)row * AZ_NSUB + sub];
            a.score_v[i] = sc;
            const bool kp = cand_keep(bx, a.min_side);
            a.keep_v[i] = kp ? 1 : 0;
            const unsigned kk = score_key(sc);
            a.key_v[i] = kp ? (kk ? kk : 1u) : 0u;
        }
        if (miss) atomicOr(&cnt->err, 8 | 256);
        if (tid == 0) { cnt->PR[l + 1] = 0; cnt->SPB[l + 1] = Un; cnt->SPN[l + 1] = 0; }
    } else if (a.lookup_next) {
        // ---- level l+1's head outputs without a head pass: every one of its regions is a child of a region of this
        //      level, and this level's pass evaluated one row per distinct RoIPool window among all such children
        //      (az_geom_dev.h: spec_children_rows).  For each unique roi of level l+1: the row of its REPRESENTATIVE
        //      (np.unique's first occurrence, test.py:214-217), raw deltas decoded against the representative's own
        //      box (test.py:241-242: `boxes = boxes[index]`), scores / zoom copied: what the tail kernel would have
        //      written for that roi, bit for bit (a roi's outputs do not depend on the launch it sits in).
        for (int i = tid; i < Un * AZ_NSUB; i += NT) {
            const int slot = i / AZ_NSUB, sub = i - slot * AZ_NSUB;
            const int rpos = sidx[slot];
            const size_t srow = (size_t)spec_base + a.crow[sprov[rpos]];
            float d4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) d4[q] = a.delta_u[srow * 4 * AZ_NSUB + 4 * sub + q];
            double bx[4];
            az_decode_box(sBn + 4 * rpos, d4, a.im_h, a.im_w, a.eps, bx);
#pragma unroll
            for (int q = 0; q < 4; ++q) a.pred_v[(size_t)i * 4 + q] = bx[q];
            const float sc = a.score_u[srow * AZ_NSUB + sub];
            a.score_v[i] = sc;
            const bool kp = cand_keep(bx, a.min_side);
            a.keep_v[i] = kp ? 1 : 0;
            const unsigned kk = score_key(sc);
            a.key_v[i] = kp ? (kk ? kk : 1u) : 0u;
        }
        for (int slot = tid; slot < Un; slot += NT)
            a.zoom_v[slot] = a.zoom_u[(size_t)spec_base + a.crow[sprov[sidx[slot]]]];
        if (tid == 0) cnt->PR[l + 1] = 0;
    } else {
        // ---- the next head pass: level l+1's unique rois, then (spec_next) one row per distinct RoIPool window among ALL
        //      children of its regions -- level l+2's outputs will be looked up there
        int S = 0;
        if (a.spec_next && l + 2 < a.nlev && Pn > 0) {
            static_assert(W_TMP2 == W_SORT + LV_C, "the window table spans the sort words and the scratch behind them");
            S = spec_children_rows<LV_C / NT>(sBn, Pn, a.scale, a.min_side, a.spatial_scale, sbuf + W_SORT, 2 * LV_C, wsum,
                                              schoff, LV_C, a.choff_next, a.crow, a.urois, a.ubox, Un, a.capR);
            if (S < 0) { if (tid == 0) atomicOr(&cnt->err, 8 | 64); return; }
        }
        if (tid == 0) { cnt->PR[l + 1] = Un + S; cnt->SPB[l + 1] = Un; cnt->SPN[l + 1] = S; }
    }
    TSTAMP();
    TREPORT();
}

__global__ void __launch_bounds__(NT) k_level_geom(AzLevelArgs a) { level_geom_body(a); }
// a batch of images searched in lockstep (az_batch.hip): workgroups (0..1, b) are image b's, its arguments in device memory
__global__ void __launch_bounds__(NT) k_level_geom_b(const AzLevelArgs *args) { AZ_UNIFORM_ARGS(AzLevelArgs, a, args + blockIdx.y); level_geom_body(a); }

}  // namespace

void azk_level_geom(hipStream_t s, const AzLevelArgs &a)
{
    static int two = -1;                 // AZ_LEVEL_WGS=1: one workgroup does both roles (measurements)
    if (two < 0) { const char *e = getenv("AZ_LEVEL_WGS"); two = (e && atoi(e) == 1) ? 0 : 1; }
    hipLaunchKernelGGL(k_level_geom, dim3(two ? 2 : 1), dim3(NT), 0, s, a);
}

void azk_level_geom_batch(hipStream_t s, const AzLevelArgs *args_dev, int n)
{
    hipLaunchKernelGGL(k_level_geom_b, dim3(2, n), dim3(NT), 0, s, args_dev);
}
