// az_fused.hip -- the first three levels of the search as two single-workgroup kernels.
//
// Levels 1-3 hold a few dozen regions (1, then the root's children, then their children), so
// each of their geometry stages is a handful of elements: as separate launches they cost ~40
// dependent kernel boundaries of pure latency.  Because the head outputs for these levels come
// from one speculative pass (az_search.hip), everything else for them -- roi keys + 1/16 dedup,
// lookup + box decode, candidate filter + ordered compaction, zoom selection, divide_region,
// _sift_dup -- runs here inside ONE workgroup with __syncthreads() between stages.
// Same device helpers (az_geom_dev.h) as the multi-workgroup kernels, so same bits.
#include "az_geom_dev.h"

namespace {

constexpr int NT = 1024;          // threads of the single workgroup
constexpr int LIM_R = 1024;       // regions per fused level

// ---- stage helpers (whole workgroup participates; all end with data visible after a sync) ----

// first-occurrence flags + rank among distinct keys (np.unique semantics) for N <= limit
// elements whose keys are in LDS.  slot[i] = position of i's key in ascending unique order.
// (Chunked dedup -- test.py:202-218 -- is expressed by the caller folding the chunk number into
// the key's top bits: equal only inside a chunk, ordered chunk-major.)
// O(N^2) LDS-broadcast compares; unrolled so the reads of several j are in flight together.
__device__ void unique_slots(const long long *skey, int N, unsigned char *sfirst, int *slot_out, int *count_out,
                             int *wsum)
{
    for (int i = threadIdx.x; i < N; i += (int)blockDim.x) {
        const long long ki = skey[i];
        int dup = 0;
#pragma unroll 8
        for (int j = 0; j < i; ++j) dup |= (skey[j] == ki);
        sfirst[i] = dup ? 0 : 1;
    }
    __syncthreads();
    int nf = 0;
    for (int i = threadIdx.x; i < N; i += (int)blockDim.x) {
        const long long ki = skey[i];
        int slot = 0;
#pragma unroll 8
        for (int j = 0; j < N; ++j) slot += (int)sfirst[j] & (int)(skey[j] < ki);
        slot_out[i] = slot;
        nf += sfirst[i];
    }
    int tot;
    block_excl_scan(nf, &tot, wsum);
    *count_out = tot;
}

}  // namespace

// ==========================================================================================
// Speculative pre-pass: B1 = divide_region(root) (with _sift_dup), all children of all of B1
// (without _sift_dup, offsets in choff_all), and the rois of S = [root ; B1 ; children].
// ==========================================================================================
__global__ void __launch_bounds__(NT)
k_spec_prepass(AzCounts *cnt, double *root, double *B1, double *child, int *choff_all,
               float *urois, double scale, double min_side, int capR, int capCh, int im_h, int im_w, int defer_root)
{
    // first kernel of a search: clear the previous search's counters and write the root region
    // (lib/detect/test.py:355) -- what k_init_root does on the multi-launch path
    {
        int *w = reinterpret_cast<int *>(cnt);
        for (int i = threadIdx.x; i < (int)(sizeof(AzCounts) / sizeof(int)); i += blockDim.x) w[i] = 0;
        if (threadIdx.x == 0) { root[0] = 0.0; root[1] = 0.0; root[2] = im_w - 1.0; root[3] = im_h - 1.0; }
        __syncthreads();
        if (threadIdx.x == 0) cnt->P[0] = 1;
        __syncthreads();
    }
    __shared__ long long skey[LIM_R];
    __shared__ unsigned char sfirst[LIM_R];
    __shared__ int sslot[LIM_R];
    __shared__ int wsum[17];
    __shared__ int s_n;
    const int tid = threadIdx.x;

    // children of the root
    const DivPlan pr = div_plan(root);
    const int n1 = div_nchildren(pr);
    if (n1 > LIM_R || n1 > capCh) { if (tid == 0) atomicOr(&cnt->err, 4); return; }
    for (int bi = tid; bi < n1; bi += NT) {
        double c[4];
        skey[bi] = div_child(root, pr, bi, min_side, c);
#pragma unroll
        for (int q = 0; q < 4; ++q) child[(size_t)bi * 4 + q] = c[q];
    }
    __syncthreads();
    int P1;
    unique_slots(skey, n1, sfirst, sslot, &P1, wsum);
    if (P1 > capR) { if (tid == 0) atomicOr(&cnt->err, 1); return; }
    for (int i = tid; i < n1; i += NT)
        if (sfirst[i]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) B1[(size_t)sslot[i] * 4 + q] = child[(size_t)i * 4 + q];
        }
    __syncthreads();

    // children of ALL of B1, no dedup
    int running = 0;
    for (int base = 0; base < P1; base += NT) {
        const int z = base + tid;
        const int n = z < P1 ? div_nchildren(div_plan(B1 + 4 * (size_t)z)) : 0;
        int tot;
        const int ex = block_excl_scan(n, &tot, wsum);
        if (z < P1) choff_all[z] = running + ex;
        running += tot;
    }
    const int CH = running;
    if (CH > capCh || 1 + P1 + CH > capR) { if (tid == 0) atomicOr(&cnt->err, 4); return; }
    __syncthreads();                              // everyone has read child[] (B1 is complete)
    for (int z = tid; z < P1; z += NT) {
        const double *r = B1 + 4 * (size_t)z;
        const DivPlan p = div_plan(r);
        const int nb = div_nchildren(p);
        const size_t o = (size_t)choff_all[z];
        for (int bi = 0; bi < nb; ++bi) {
            double c[4];
            div_child(r, p, bi, min_side, c);
#pragma unroll
            for (int q = 0; q < 4; ++q) child[(o + bi) * 4 + q] = c[q];
        }
    }
    __syncthreads();
    // (defer_root: the root's row rides on the level-4 launch instead -- its zoom is forced and its candidates
    //  are only needed by the final selection --, which leaves 48 rows = 1.5 strips here for a 600x1000 image)
    const int rb = defer_root ? 0 : 1;
    const int total = rb + P1 + CH;
    for (int i = tid; i < total; i += NT) {
        const int j = i + 1 - rb;                          // row of the undeferred layout
        const double *b = (j == 0) ? root : (j <= P1 ? B1 + 4 * (size_t)(j - 1) : child + 4 * (size_t)(j - 1 - P1));
        urois[5 * (size_t)i] = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) urois[5 * (size_t)i + 1 + q] = (float)(b[q] * scale);
    }
    if (tid == 0) { cnt->specP1 = P1; cnt->specCH = CH; cnt->specU = total; }
    (void)s_n;
}

// ==========================================================================================
// Levels 0 .. n_fused-1 of the search loop (lib/detect/test.py:373-391), head outputs looked up
// in the speculative pass (zoom_s / score_s / delta_s).  All per-level state (regions, keys,
// dedup slots, zoom set, child keys) lives in LDS; global memory is touched only to read the
// speculative head outputs and to append candidates, so a level costs one memory round trip
// instead of one per stage.  If a level outgrows the LDS tables the kernel reports bit 3 of
// cnt->err and the host reruns the search with the multi-launch kernels.
// ==========================================================================================
constexpr int FL_R = 256;         // regions per fused level (LDS-resident)
constexpr int FL_C = 2048;        // children per fused level

#ifdef AZ_FUSED_TIMING
#define TSTAMP(i) do { __syncthreads(); if (tid == 0) ts[i] = wall_clock64(); } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif
constexpr int SPEC_PRE = 64;     // rows of the speculative pass whose outputs are staged in LDS up front
#ifndef AZ_NTL
#define AZ_NTL 512      /* (128: 61 us, 256: 43.5, 512: 40, 1024: 41 -- k_spec_levels at config A) */
#endif
constexpr int NTL = AZ_NTL;         // threads of the fused-levels workgroup: levels 1-3 hold <= a few hundred elements per stage,
                                 // and every stage boundary costs a barrier across all waves
static __device__ __forceinline__ void spec_levels_body(const AzFusedArgs &a)
{
    __shared__ double sB[2][FL_R * 4];
    __shared__ int ssrc[2][FL_R];
    __shared__ long long skey[FL_R];
    // (child keys of the level being divided; behind them the scratch that, together, is the window table of the
    //  pair-speculation stage: 2 * FL_C contiguous words)
    __shared__ unsigned long long stable[2 * FL_C];
    long long *skeyC = reinterpret_cast<long long *>(stable);
    __shared__ unsigned char sfirst[FL_C];
    __shared__ int sslot[FL_C];
    __shared__ int sczi[FL_C];
    __shared__ int sidx[FL_R], szr[FL_R], schoff[FL_R + 1];
    __shared__ unsigned long long ssort[2 * FL_R];
    __shared__ unsigned sbins[SORT_NB + 1];
    __shared__ unsigned smm[2];
    __shared__ int wsum[17];
    const int tid = threadIdx.x;
    AzCounts *cnt = a.cnt;
#ifdef AZ_FUSED_TIMING
    __shared__ unsigned long long ts[64];
    int tsn = 0;
#endif
    TSTAMP(tsn++);

    if (a.reset) {
        // first kernel of the search that touches the counters (the pre-pass of this image shape is cached,
        // az_search.hip): clear the previous search's, restore what the pre-pass would have left
        int *w = reinterpret_cast<int *>(cnt);
        for (int i = tid; i < (int)(sizeof(AzCounts) / sizeof(int)); i += NTL) w[i] = 0;
        __syncthreads();
        if (tid == 0) { cnt->P[0] = 1; cnt->specP1 = a.specP1; cnt->specCH = a.specCH; cnt->specU = a.specU; }
    }
    if (tid == 0) {                                  // lib/detect/test.py:355
        sB[0][0] = 0.0; sB[0][1] = 0.0; sB[0][2] = a.im_w - 1.0; sB[0][3] = a.im_h - 1.0;
    }
    // The speculative pass's outputs (a few dozen rows) and the pre-pass's child offsets come in with ONE round trip:
    // every stage below then reads LDS instead of paying a dependent global load (~1-2 us each, a dozen times).
    __shared__ float s_zs[SPEC_PRE], s_ss[SPEC_PRE * AZ_NSUB], s_ds[SPEC_PRE * 4 * AZ_NSUB];
    __shared__ int s_choff[SPEC_PRE];
    const int specU_h = a.reset ? a.specU : cnt->specU;
    const int P1spec = a.reset ? a.specP1 : cnt->specP1;
    const bool pre = specU_h <= SPEC_PRE && P1spec <= SPEC_PRE;
    if (a.stab && !pre) { if (tid == 0) atomicOr(&cnt->err, 8 | 256); return; }       // (whole-tree pass: rows come through the map)
    if (pre) {
        if (a.stab) {
            // whole-tree speculation: row i of the speculative layout is row row_map[i] of the one pass
            auto rowof = [&](int i) { const int m = a.row_map[i]; return m == (int)AZ_TAB_ROOT ? a.root_row : m; };
            for (int i = tid; i < specU_h; i += NTL) s_zs[i] = a.zoom_s[rowof(i)];
            for (int i = tid; i < specU_h * AZ_NSUB; i += NTL) {
                const int r = i / AZ_NSUB;
                s_ss[i] = a.score_s[(size_t)rowof(r) * AZ_NSUB + (i - r * AZ_NSUB)];
            }
            for (int i = tid; i < specU_h * 4 * AZ_NSUB; i += NTL) {
                const int r = i / (4 * AZ_NSUB);
                s_ds[i] = a.delta_s[(size_t)rowof(r) * 4 * AZ_NSUB + (i - r * 4 * AZ_NSUB)];
            }
        } else {
            for (int i = tid; i < specU_h; i += NTL) s_zs[i] = a.zoom_s[i];
            for (int i = tid; i < specU_h * AZ_NSUB; i += NTL) s_ss[i] = a.score_s[i];
            for (int i = tid; i < specU_h * 4 * AZ_NSUB; i += NTL) s_ds[i] = a.delta_s[i];
        }
        for (int i = tid; i < P1spec; i += NTL) s_choff[i] = a.choff_all[i];
    }
    // (deferred root: level 2's regions are the pre-pass's B1 -- same round trip)
    if (a.defer_root && P1spec <= FL_R)
        for (int i = tid; i < P1spec * 4; i += NTL) sB[1][i] = a.specB1[i];
    const float *zoom_s = pre ? s_zs : a.zoom_s, *score_s = pre ? s_ss : a.score_s, *delta_s = pre ? s_ds : a.delta_s;
    const int *choff_all = pre ? s_choff : a.choff_all;
    __syncthreads();
    int P = 1;
    int ybase = 0;
    int l0 = 0;
    if (a.defer_root) {
        // Level 1 needs no head output here: the root's zoom is forced (test.py:383-384), its candidates arrive
        // later (11 reserved slots), and its children after _sift_dup are the pre-pass's B1.
        const bool rz = (1.0 >= a.Tz);                  // zoom[0] = 1, then indZ = where(zoom >= Tz)
        P = rz ? P1spec : 0;
        if (P > FL_R) { if (tid == 0) atomicOr(&cnt->err, 8); return; }
        if (tid == 0) {
            cnt->P[0] = 1; cnt->U[0] = 1; cnt->NC[0] = AZ_NSUB; cnt->ytot[0] = 0; cnt->PZ[0] = rz ? 1 : 0;
            cnt->CH[0] = rz ? div_nchildren(div_plan(sB[0])) : 0;
        }
        ybase = AZ_NSUB;
        l0 = 1;
        __syncthreads();
    }
    for (int l = l0; l < a.n_fused; ++l) {
        const int cur = l & 1;
        const double *B = sB[cur];
        if (tid == 0) { cnt->P[l] = P; cnt->ytot[l] = ybase; }
        if (a.cut_short && l == 2 && P > 0) {        // the pass holds no rows for this level: the host runs the search again
            if (tid == 0) atomicOr(&cnt->err, 1024);
            return;
        }
        if (P == 0) {                                // Z was empty: the reference's loop breaks
            if (tid == 0)
                for (int ll = l; ll < a.n_fused; ++ll) { cnt->P[ll] = 0; cnt->ytot[ll + 1] = ybase; }
            break;                                   // (the closing stage still runs: a deferred root needs its row)
        }

        // ---- roi projection + feature-space dedup (test.py:61-97, 210-218) -------------------
        for (int r = tid; r < P; r += NTL) {
            float roi5[5];
            // roi keys are < 1000^5 < 2^50: the dedup chunk number (test.py:202-205) rides above them
            skey[r] = roi_and_key(B + 4 * r, a.scale, a.dedup, roi5, r) + ((long long)(r / a.batch) << 50);
        }
        __syncthreads();
        TSTAMP(tsn++);
        int U;
        unique_slots(skey, P, sfirst, sslot, &U, wsum);
        for (int i = tid; i < P; i += NTL)
            if (sfirst[i]) sidx[sslot[i]] = i;       // index[]: representative of each unique roi
        __syncthreads();
        // speculative row of region r's representative (level 0: the root; 1: 1 + index; 2: carried)
        const int rb = a.defer_root ? 0 : 1;          // rows before B1's in the speculative pass
        auto spec_row = [&](int r, int &rep) {
            rep = sidx[sslot[r]];
            return l == 0 ? 0 : (l == 1 ? rb + rep : ssrc[cur][rep]);
        };

        TSTAMP(tsn++);
        // ---- candidates: decode + clip against the representative's box (test.py:106-151), filter,
        //      ordered append to Y / aScores (test.py:171-187, 380-381) ------------------------------
        int run = 0;
        const bool deferred = a.defer_root && l == 0;      // the root's candidates arrive with level 4's head pass:
        if (deferred) run = AZ_NSUB;                        // their slots are reserved (az_level.hip fills / closes them)
        for (int base = 0; base < P * AZ_NSUB && !deferred; base += NTL) {
            const int c = base + tid;
            int fl = 0;
            double bx[4];
            float sc = 0.f;
            if (c < P * AZ_NSUB) {
                const int r = c / AZ_NSUB, s = c - r * AZ_NSUB;
                int rep;
                const int srow = spec_row(r, rep);
                float d4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) d4[q] = delta_s[(size_t)srow * 4 * AZ_NSUB + 4 * s + q];
                az_decode_box(B + 4 * rep, d4, a.im_h, a.im_w, a.eps, bx);
                fl = cand_keep(bx, a.min_side);
                sc = score_s[(size_t)srow * AZ_NSUB + s];
            }
            int tot;
            const int off = block_excl_scan(fl, &tot, wsum);
            const int dst = ybase + run + off;
            if (fl && dst < a.capCand) {
#pragma unroll
                for (int q = 0; q < 4; ++q) a.Yall[(size_t)dst * 4 + q] = bx[q];
                a.Sall[dst] = sc;
            }
            run += tot;
        }
        int nc = run;
        if (ybase + nc > a.capCand) { nc = a.capCand - ybase; if (tid == 0) atomicOr(&cnt->err, 2); }
        TSTAMP(tsn++);
        // ---- zoom selection (test.py:383-387) -------------------------------------------------------
        int PZ = 0;
        for (int base = 0; base < P; base += NTL) {
            const int r = base + tid;
            int zf = 0;
            if (r < P) {
                int rep;
                float z = 1.0f;                                   // test.py:383-384: zoom[0] = 1 at level 1
                if (!(l == 0 && r == 0)) z = zoom_s[spec_row(r, rep)];
                zf = ((double)z >= a.Tz);
            }
            int tot;
            const int off = block_excl_scan(zf, &tot, wsum);
            if (zf) szr[PZ + off] = r;
            PZ += tot;
        }
        if (tid == 0) { cnt->U[l] = U; cnt->NC[l] = nc; cnt->ytot[l + 1] = ybase + nc; cnt->PZ[l] = PZ; }
        ybase += nc;
        __syncthreads();
        if (l + 1 >= a.nlev) return;

        TSTAMP(tsn++);
        // ---- divide_region + _sift_dup (div.pyx:15-89) ---------------------------------------------
        int CH = 0;
        for (int base = 0; base < PZ; base += NTL) {
            const int z = base + tid;
            const int n = z < PZ ? div_nchildren(div_plan(B + 4 * szr[z])) : 0;
            int tot;
            const int ex = block_excl_scan(n, &tot, wsum);
            if (z < PZ) {
                schoff[z] = CH + ex;
                if (CH + ex + n <= FL_C)
                    for (int bi = 0; bi < n; ++bi) sczi[CH + ex + bi] = (z << 16) | bi;      // (child -> parent, child number)
            }
            CH += tot;
        }
        if (CH > FL_C || CH > a.capCh) { if (tid == 0) atomicOr(&cnt->err, 8); return; }
        __syncthreads();
        // (one thread per CHILD: a parent's children one after the other are ~35 dependent f64 divisions)
        for (int ci = tid; ci < CH; ci += NTL) {
            const int z = sczi[ci] >> 16;
            const double *r = B + 4 * szr[z];
            const int bi = sczi[ci] & 0xFFFF;
            double c[4];
            skeyC[ci] = div_child(r, div_plan(r), bi, a.min_side, c);
        }
        __syncthreads();
        TSTAMP(tsn++);
        int Pn;
        unique_slots(skeyC, CH, sfirst, sslot, &Pn, wsum);
        TSTAMP(tsn++);
        if (Pn > FL_R || Pn > a.capR) { if (tid == 0) atomicOr(&cnt->err, 8); return; }
        for (int i = tid; i < CH; i += NTL)
            if (sfirst[i]) {
                const int slot = sslot[i];
                const int z = sczi[i] >> 16, bi = sczi[i] & 0xFFFF;
                const int pr = szr[z];
                const double *r = B + 4 * pr;
                div_child(r, div_plan(r), bi, a.min_side, &sB[cur ^ 1][4 * slot]);
                // level-3 regions remember their row in the speculative pass: (parent in B1, child)
                ssrc[cur ^ 1][slot] = (l == 1) ? rb + P1spec + choff_all[pr] + bi : 0;
            }
        if (tid == 0) cnt->CH[l] = CH;
        P = Pn;
        __syncthreads();
    }
    TSTAMP(tsn++);
#ifdef AZ_FUSED_TIMING
    if (tid == 0) {
        printf("spec_levels stamps (x10ns):");
        for (int i = 1; i < tsn; ++i) printf(" %llu", ts[i] - ts[i - 1]);
        printf("\n");
    }
#endif
    // hand the next level's regions to the multi-workgroup kernels
    const int nxt = a.n_fused & 1;
#ifdef AZ_SPEC_TIMING
    __syncthreads();
    const unsigned long long tc0 = wall_clock64();
#endif
    for (int i = tid; i < P * 4; i += NTL) a.B[nxt][i] = sB[nxt][i];
    if (tid == 0) cnt->P[a.n_fused] = P;
    if (a.cut_next) {                // the host stopped the search here: right if the tree did, too
        if (tid == 0 && P > 0) atomicOr(&cnt->err, 1024);
        return;
    }
    if (a.next_dedup && a.n_fused < a.nlev) {
        // the next level runs on the fused level kernel (az_level.hip), which expects its rois deduplicated
        if (P > a.batch) { if (tid == 0) atomicOr(&cnt->err, 8); return; }      // chunked dedup: multi-launch path
        __syncthreads();
        const int U = roi_dedup_sorted(sB[nxt], P, a.scale, a.dedup, ssort, ssort + FL_R, sbins, smm, wsum, nullptr,
                                       a.index, a.inv, a.urois, a.ubox, sidx);
        if (a.stab) {
            // whole-tree speculation: that level's head outputs by RoIPool window among the rows of the one pass (what
            // k_level_geom's lookup stage does for the levels after it): raw deltas decoded against the representative's own
            // box, scores / zoom copied -- what the tail kernel would have written for that roi, bit for bit
            __syncthreads();
            // (one probe per unique roi, its row parked in LDS; then one thread per (roi, sub-region) decodes)
            int miss = 0;
            for (int slot = tid; slot < U; slot += NTL) {
                const double *bx0 = sB[nxt] + 4 * sidx[slot];
                float roi5[5];
                roi5[0] = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q) roi5[1 + q] = (float)(bx0[q] * a.scale);
                const int row = az_tab_lookup(a.stab, a.stabT, roi5, a.spatial_scale, a.root_row);
                sslot[slot] = row;
                if (row < 0) miss = 1; else a.zoom_v[slot] = a.zoom_s[row];
            }
            __syncthreads();
            for (int i = tid; i < U * AZ_NSUB; i += NTL) {
                const int slot = i / AZ_NSUB, sub = i - slot * AZ_NSUB;
                const int row = sslot[slot];
                if (row < 0) continue;
                const double *bx0 = sB[nxt] + 4 * sidx[slot];
                float d4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) d4[q] = a.delta_s[(size_t)row * 4 * AZ_NSUB + 4 * sub + q];
                double bx[4];
                az_decode_box(bx0, d4, a.im_h, a.im_w, a.eps, bx);
#pragma unroll
                for (int q = 0; q < 4; ++q) a.pred_v[(size_t)i * 4 + q] = bx[q];
                const float sc = a.score_s[(size_t)row * AZ_NSUB + sub];
                a.score_v[i] = sc;
                const bool kp = cand_keep(bx, a.min_side);
                a.keep_v[i] = kp ? 1 : 0;
                const unsigned kk = score_key(sc);
                a.key_v[i] = kp ? (kk ? kk : 1u) : 0u;
            }
            if (miss) atomicOr(&cnt->err, 8 | 256);
            if (tid == 0) { cnt->U[a.n_fused] = U; cnt->SPB[a.n_fused] = U; cnt->SPN[a.n_fused] = 0; cnt->PR[a.n_fused] = 0; }
            return;
        }
#ifdef AZ_SPEC_TIMING
        __syncthreads();
        if (tid == 0) printf("closing: handover + roi dedup %llu (x10ns)\n", wall_clock64() - tc0);
#endif
        // pair speculation: that level's head pass also evaluates one row per distinct RoIPool window among ALL
        // children of its regions (a superset of the level after it), behind its own unique rois
        int S = 0;
        if (a.spec_next && a.n_fused + 1 < a.nlev && P > 0) {
            __syncthreads();
            S = spec_children_rows<FL_C / NTL>(sB[nxt], P, a.scale, a.min_side, a.spatial_scale, stable, 2 * FL_C, wsum,
                                               schoff, FL_C, a.choff_next, a.crow, a.urois, a.ubox, U, a.capR - 1);
            if (S < 0) { if (tid == 0) atomicOr(&cnt->err, 8 | 64); return; }
        }
        if (tid == 0) {
            cnt->U[a.n_fused] = U;
            int rows = U + S;                          // rows the head evaluates at that level
            cnt->SPB[a.n_fused] = U;
            cnt->SPN[a.n_fused] = S;
            if (a.defer_root) {
                // the deferred root rides as the LAST row (RoIPool treats the tail of a launch cooperatively)
                const double rootb[4] = {0.0, 0.0, a.im_w - 1.0, a.im_h - 1.0};
                a.urois[5 * (size_t)rows] = 0.0f;
                for (int q = 0; q < 4; ++q) {
                    a.urois[5 * (size_t)rows + 1 + q] = (float)(rootb[q] * a.scale);
                    a.ubox[4 * (size_t)rows + q] = rootb[q];
                }
                ++rows;
            }
            cnt->PR[a.n_fused] = rows;
        }
    }
}

__global__ void __launch_bounds__(NTL) k_spec_levels(AzFusedArgs a) { spec_levels_body(a); }
// a batch of images searched in lockstep (az_batch.hip): workgroup (0, b) is image b's, its arguments in device memory
__global__ void __launch_bounds__(NTL) k_spec_levels_b(const AzFusedArgs *args) { AZ_UNIFORM_ARGS(AzFusedArgs, a, args + blockIdx.y); spec_levels_body(a); }

// ------------------------------------------------------------------------------------------
void azk_spec_prepass(hipStream_t s, AzCounts *cnt, double *root, double *B1, double *child, int *choff_all,
                      float *urois, double scale, double min_side, int capR, int capCh, int im_h, int im_w,
                      int defer_root)
{
    hipLaunchKernelGGL(k_spec_prepass, dim3(1), dim3(NT), 0, s, cnt, root, B1, child, choff_all, urois, scale,
                       min_side, capR, capCh, im_h, im_w, defer_root);
}

void azk_spec_levels(hipStream_t s, const AzFusedArgs &a)
{
    hipLaunchKernelGGL(k_spec_levels, dim3(1), dim3(NTL), 0, s, a);
}

void azk_spec_levels_batch(hipStream_t s, const AzFusedArgs *args_dev, int n)
{
    hipLaunchKernelGGL(k_spec_levels_b, dim3(1, n), dim3(NTL), 0, s, args_dev);
}
