// az_geom_dev.h -- device helpers shared by the multi-workgroup geometry kernels (az_geom.hip)
// and the single-workgroup fused kernels of the first levels (az_fused.hip).  All f64/f32
// arithmetic here is written in the reference's operation order and must be compiled with
// -ffp-contract=off.
#pragma once
#include "az_dev.h"

static __device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Inclusive prefix sum across the 64 lanes of a wave with DPP row shifts / row broadcasts (six VALU instructions with a
// cross-lane modifier; the __shfl_up form is six dependent ds_bpermute round trips through the LDS crossbar).
static __device__ __forceinline__ int wave_incl_scan(int v)
{
#ifdef AZ_SCAN_SHFL
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
#else
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);     // row_shr:8   (each row of 16 lanes scanned)
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);     // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);     // row_bcast:31 -> rows 2 and 3
    return v;
#endif
}

// min / max across a wave, same DPP ladder (the value of lane 63 is the wave's)
static __device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
#define AZ_DPP_STEP(ctrl, rmask) { const unsigned t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xF, false); v = t < v ? t : v; }
    AZ_DPP_STEP(0x111, 0xF) AZ_DPP_STEP(0x112, 0xF) AZ_DPP_STEP(0x114, 0xF) AZ_DPP_STEP(0x118, 0xF)
    AZ_DPP_STEP(0x142, 0xA) AZ_DPP_STEP(0x143, 0xC)
#undef AZ_DPP_STEP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
static __device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
#define AZ_DPP_STEP(ctrl, rmask) { const unsigned t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rmask, 0xF, false); v = t > v ? t : v; }
    AZ_DPP_STEP(0x111, 0xF) AZ_DPP_STEP(0x112, 0xF) AZ_DPP_STEP(0x114, 0xF) AZ_DPP_STEP(0x118, 0xF)
    AZ_DPP_STEP(0x142, 0xA) AZ_DPP_STEP(0x143, 0xC)
#undef AZ_DPP_STEP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Exclusive prefix sum of one int per thread across the block (blockDim.x <= 1024).
// Two barriers; every wave scans the (<= 16) wave totals itself with the same DPP scan -- no thread walks them one
// dependent LDS access after the other.  (The single-workgroup geometry kernels make dozens of these calls per search:
// at ~1 us each in the __shfl_up + serial-walk form they were most of those kernels' time.)
static __device__ int block_excl_scan(int v, int *total, int *wsum /* >= 17 ints of LDS */)
{
    const int lane = lane_id(), nw = (blockDim.x + 63) >> 6;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int inc = wave_incl_scan(v);
    __syncthreads();                       // wsum may still be read from a previous call
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
#ifdef AZ_SCAN_OLD
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < nw; ++w) { int t = wsum[w]; wsum[w] = run; run += t; }
        wsum[16] = run;
    }
    __syncthreads();
    *total = wsum[16];
    return wsum[wid] + inc - v;
#else
    const int t = lane < nw ? wsum[lane] : 0;
    const int tinc = wave_incl_scan(t);
    *total = __builtin_amdgcn_readlane(tinc, 63);
    return __builtin_amdgcn_readlane(tinc - t, wid) + inc - v;
#endif
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


// ----------------------------------------------------------------------------------------
// Single-workgroup building blocks of the fused level kernels (az_fused.hip, az_level.hip).

// Sort (ascending) of N DISTINCT 64-bit words in LDS by the whole workgroup, as a bucket sort: the words'
// high parts (w >> S, which must fit 32 bits) are mapped linearly from [min, max] onto SORT_NB buckets
// (monotone, so bucket order is sort order), a histogram + scan gives every bucket its segment, and inside a
// segment -- a handful of words for the region hashes and scores of this path -- a word's place is the number
// of smaller words.  Seven barriers and no dependent compare-exchange chain (a bitonic network of 1024 words
// took 9 us on one CU, this takes ~2); correct for any input, O(N^2 / threads) only if all words share a
// bucket.  `tmp`: N words, `bins`: SORT_NB + 1 counters, `mm`: 2 words of LDS.
constexpr int SORT_NB = 4096;

static __device__ void block_bucket_sort(unsigned long long *w, int N, unsigned long long *tmp, unsigned *bins,
                                         int S, int *wsum, unsigned *mm)
{
    const int tid = threadIdx.x, nt = (int)blockDim.x;
    // buckets in use: ~4 per word (a power of two, at most SORT_NB) -- clearing and scanning 4096 counters for a few
    // hundred words was most of the sort's time
    int NB = 256;
    while (NB < 4 * N && NB < SORT_NB) NB <<= 1;
    __syncthreads();
    if (tid == 0) { mm[0] = 0xFFFFFFFFu; mm[1] = 0u; }
    for (int b = tid; b <= NB; b += nt) bins[b] = 0u;
    __syncthreads();
    unsigned lo = 0xFFFFFFFFu, hi = 0u;
    for (int i = tid; i < N; i += nt) {
        const unsigned h = (unsigned)(w[i] >> S);
        lo = h < lo ? h : lo;
        hi = h > hi ? h : hi;
    }
    lo = wave_min_u32(lo);
    hi = wave_max_u32(hi);
    if ((tid & 63) == 0 && N > 0) { atomicMin(&mm[0], lo); atomicMax(&mm[1], hi); }
    __syncthreads();
    const unsigned base_h = mm[0];
    const unsigned long long range = (unsigned long long)(mm[1] - mm[0]) + 1ull;
    auto bucket = [&](unsigned long long x) {
        return (int)(((unsigned long long)((unsigned)(x >> S) - base_h) * (unsigned long long)NB) / range);
    };
    for (int i = tid; i < N; i += nt) atomicAdd(&bins[bucket(w[i])], 1u);
    __syncthreads();
    {   // exclusive scan of the bucket counts, in place
        const int per = (NB + nt - 1) / nt;
        const int b0 = min(NB, tid * per), b1 = min(NB, b0 + per);
        int sum = 0;
        for (int b = b0; b < b1; ++b) sum += (int)bins[b];
        int tot;
        int run = block_excl_scan(sum, &tot, wsum);
        for (int b = b0; b < b1; ++b) { const int c = (int)bins[b]; bins[b] = (unsigned)run; run += c; }
    }
    __syncthreads();
    // scatter into the bucket segments (arrival order inside a segment); bins[b] ends up as the END of segment b
    for (int i = tid; i < N; i += nt) {
        const unsigned long long x = w[i];
        tmp[atomicAdd(&bins[bucket(x)], 1u)] = x;
    }
    __syncthreads();
    for (int p = tid; p < N; p += nt) {
        const unsigned long long x = tmp[p];
        const int b = bucket(x);
        const int s0 = b ? (int)bins[b - 1] : 0, s1 = (int)bins[b];
        int r = 0;
        for (int q = s0; q < s1; ++q) r += tmp[q] < x;
        w[s0 + r] = x;
    }
    __syncthreads();
}

// Segment of position i in an ascending offset table off[0 .. n): the last s with off[s] <= i.
static __device__ __forceinline__ int seg_of(const int *off, int n, int i)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

static __device__ __forceinline__ int next_pow2(int v)
{
    int n = 2;
    while (n < v) n <<= 1;
    return n;
}

// Roi projection + feature-space dedup of the Pn regions `Bn` (LDS or global; lib/detect/test.py:61-97,
// 210-218) by one workgroup: rois [Pn,5] (written only if `rois` is not NULL), and np.unique(hashes, return_index, return_inverse) as a sort of
// (key << 14 | position): run heads in ascending key order are the unique rois, the head of a run is its first
// occurrence.  Writes index / inv / the unique rois and their anchor boxes, returns U.  The caller guarantees
// Pn <= 16384 and Pn <= batch (one dedup chunk: keys < 1000^5 < 2^50); `ssort` / `stmp` hold Pn words each.
static __device__ int roi_dedup_sorted(const double *Bn, int Pn, double scale, float dedup, unsigned long long *ssort,
                                       unsigned long long *stmp, unsigned *bins, unsigned *mm, int *wsum, float *rois,
                                       int *index, int *inv, float *urois, double *ubox, int *sidx = nullptr)
{
    const int tid = threadIdx.x, nt = (int)blockDim.x;
    for (int r = tid; r < Pn; r += nt) {
        float roi5[5];
        const long long key = roi_and_key(Bn + 4 * r, scale, dedup, roi5, r);
        if (rois) {
#pragma unroll
            for (int q = 0; q < 5; ++q) rois[5 * (size_t)r + q] = roi5[q];
        }
        ssort[r] = ((unsigned long long)key << 14) | (unsigned)r;
    }
    block_bucket_sort(ssort, Pn, stmp, bins, 34, wsum, mm);            // high part = key >> 20
    int U = 0;
    for (int base = 0; base < Pn; base += nt) {
        const int i = base + tid;
        int head = 0;
        unsigned long long w = 0;
        if (i < Pn) {
            w = ssort[i];
            head = (i == 0) || ((ssort[i - 1] >> 14) != (w >> 14));
        }
        int tot;
        const int ex = block_excl_scan(head, &tot, wsum);
        if (i < Pn) {
            const int slot = U + ex + head - 1;            // run number of position i
            const int r = (int)(w & 0x3FFFu);
            inv[r] = slot;
            if (head) {
                index[slot] = r;
                if (sidx) sidx[slot] = r;                                // (LDS copy for the caller's next stage)
                float roi5[5];
                roi_and_key(Bn + 4 * r, scale, dedup, roi5, r);          // (recomputed: cheaper than a memory round trip)
#pragma unroll
                for (int q = 0; q < 5; ++q) urois[5 * (size_t)slot + q] = roi5[q];
#pragma unroll
                for (int q = 0; q < 4; ++q) ubox[4 * (size_t)slot + q] = Bn[4 * r + q];
            }
        }
        U += tot;
    }
    return U;
}

// ----------------------------------------------------------------------------------------
// Pair speculation (az_search.hip): rows for ALL children of all P regions `Bn` of a level, appended to that level's
// head pass.  The next level's regions are _sift_dup(divide_region(Z)) with Z a SUBSET of these parents
// (test.py:386-390), so every region the next level can hold is one of these children, bit for bit (a child is a
// function of its parent alone).  The head's outputs for a roi (zoom, scores, raw deltas) depend on the roi only
// through RoIPool's integer window -- C round() of coordinate * spatial_scale (ROIPooling, test_fc.prototxt:14-25) --,
// so children are deduplicated by those four integers: one row per distinct window, whichever child supplies the
// coordinates.  Which child REPRESENTS a roi in the reference's own np.round dedup, and the anchor box its deltas are
// decoded against, are decided later from the real tree (az_level.hip: lookup stage); here only: child ci -> row.
//   choff_all_g[r]  first child of region r in the all-children list (global, P ints; also `schoff` in LDS, P ints)
//   crow_g[ci]      spec row (0-based among the spec rows) of child ci (global)
//   urois / ubox    the spec rows' rois (f32, scaled) and, as a placeholder anchor, the supplying child's box, written at
//                   rows row_base .. row_base + S
// Returns S, or -1 if the children outgrow maxC / the rows outgrow capRows / a coordinate leaves the key's range.
static __device__ __forceinline__ bool pool_key(const float *roi5, float ss, unsigned long long *key)
{
    unsigned long long k = 0;
    bool ok = true;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int v = (int)roundf(roi5[1 + q] * ss) + 2048;          // k_roi_pool's own expression
        ok = ok && v >= 0 && v < 4096;
        k |= (unsigned long long)(v & 4095) << (12 * q);
    }
    *key = k;
    return ok;
}

// Whole-tree speculation (az_search.hip: SearchPlan::full): ONE head pass evaluates the unique rois of the image shape's full
// tree (the one-pass plan's rows); a search with any Tz then finds a region's head outputs by its RoIPool window in a
// table built once per shape: open addressing in global memory, word = (window key << 13) | row, EMPTY = ~0.
// Row AZ_TAB_ROOT stands for the root's row (the last row of the pass, wherever the extra rows push it).
constexpr unsigned AZ_TAB_ROOT = 0x1FFFu;
static __device__ __forceinline__ unsigned az_tab_hash(unsigned long long key, unsigned T)
{
    return (unsigned)((((key * 0x9E3779B97F4A7C15ull) >> 32) * (unsigned long long)T) >> 32);
}
// -1: the window is not in the table
static __device__ __forceinline__ int az_tab_lookup(const unsigned long long *tab, unsigned T, const float *roi5, float ss,
                                                    int root_row)
{
    unsigned long long key;
    if (!pool_key(roi5, ss, &key)) return -1;
    unsigned slot = az_tab_hash(key, T);
    for (unsigned probes = 0; probes < T; ++probes) {
        const unsigned long long w = tab[slot];
        if (w == ~0ull) return -1;
        if ((w >> 13) == key) { const unsigned r = (unsigned)(w & 0x1FFFu); return r == AZ_TAB_ROOT ? root_row : (int)r; }
        slot = slot + 1 == T ? 0u : slot + 1;
    }
    return -1;
}

// one child's box without its _sift_dup hash (the hash costs four f64 divisions nobody needs here)
static __device__ __forceinline__ void div_child_box(const double *r, const DivPlan &p, int bi, double *c)
{
    const double h_short = p.l_short / 2, h_long = p.l_long / 2;   // div.pyx:58-59
    double s_lo, s_hi, l_lo, l_hi;
    if (bi < (int)(2 * p.num_long)) {
        const unsigned k = (unsigned)bi / p.num_long, j = (unsigned)bi - k * p.num_long;
        s_lo = k * p.l_short; s_hi = (k + 1) * p.l_short;
        l_lo = j * p.l_long;  l_hi = (j + 1) * p.l_long;
    } else {
        const unsigned j = (unsigned)bi - 2 * p.num_long;
        s_lo = 0 * p.l_short + h_short; s_hi = (0 + 1) * p.l_short + h_short;
        l_lo = j * p.l_long + h_long;   l_hi = (j + 1) * p.l_long + h_long;
    }
    if (p.min_ind == 0) { c[0] = s_lo; c[1] = l_lo; c[2] = s_hi; c[3] = l_hi; }
    else                { c[0] = l_lo; c[1] = s_lo; c[2] = l_hi; c[3] = s_hi; }
    c[0] += r[0]; c[2] += r[0]; c[1] += r[1]; c[3] += r[1];        // div.pyx:71-72
}

// Dedup by RoIPool window = an open-addressing table in LDS, one 64-bit word per entry: (window key << 13) | child.
// A slot is claimed for a key by compare-and-swap on the empty word and then keeps the SMALLEST child index of that
// key (atomicMin on the whole word: same key, so the order is the child's) -- the representative is therefore the first
// child of the window in all-children order, whatever the order the threads arrive in.  Rows are numbered in that
// order too (a block scan over the head flags), so the row layout is deterministic.
//   buf: W words of LDS: the table (3W/4 words) and, behind it, one int per child (its slot; W/2 ints >= maxC)
//   MAXIT >= ceil(maxC / blockDim): each thread keeps the boxes of its children in registers between the two phases
template <int MAXIT>
static __device__ int spec_children_rows(const double *Bn, int P, double scale, double min_side, float ss,
                                         unsigned long long *buf, int W, int *wsum, int *schoff, int maxC,
                                         int *choff_all_g, int *crow_g, float *urois, double *ubox, int row_base,
                                         int capRows)
{
    (void)min_side;
    const int tid = threadIdx.x, nt = (int)blockDim.x;
    constexpr unsigned long long EMPTY = ~0ull;
    const unsigned T = (unsigned)(W / 4 * 3);
    unsigned long long *tab = buf;
    int *sslot = reinterpret_cast<int *>(buf + T);
#ifdef AZ_SPEC_TIMING
    unsigned long long tq[8]; int tqn = 0;
#define SPEC_T() do { __syncthreads(); if (tqn < 8) tq[tqn++] = wall_clock64(); } while (0)
#else
#define SPEC_T() do { } while (0)
#endif
    SPEC_T();
    for (unsigned i = tid; i < T; i += nt) tab[i] = EMPTY;
    int CH = 0;
    for (int base = 0; base < P; base += nt) {
        const int z = base + tid;
        const int n = z < P ? div_nchildren(div_plan(Bn + 4 * z)) : 0;
        int tot;
        const int o = CH + block_excl_scan(n, &tot, wsum);
        if (z < P) {
            schoff[z] = o; choff_all_g[z] = o;
            // (every child's parent, written by the parent's thread: the child threads below read it instead of
            //  searching the offsets -- eight dependent LDS reads each)
            if (o + n <= maxC && o + n <= W / 2)
                for (int bi = 0; bi < n; ++bi) sslot[o + bi] = z;
        }
        CH += tot;
    }
    if (CH > maxC || CH > 8192 || CH > W / 2 || CH > MAXIT * nt) return -1;
    __syncthreads();
    SPEC_T();
    int bad = 0;
    double cb[MAXIT][4];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int ci = tid + it * nt;                                    // one thread per child
        if (ci < CH) {
            const int r = sslot[ci];
            div_child_box(Bn + 4 * r, div_plan(Bn + 4 * r), ci - schoff[r], cb[it]);
            float roi5[5];
            roi5[0] = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) roi5[1 + q] = (float)(cb[it][q] * scale);   // test.py:61-97
            unsigned long long key;
            if (!pool_key(roi5, ss, &key)) bad = 1;
            const unsigned long long word = (key << 13) | (unsigned)ci;
            // multiplicative hash of the 48-bit key onto [0, T)
            unsigned slot = (unsigned)((((key * 0x9E3779B97F4A7C15ull) >> 32) * (unsigned long long)T) >> 32);
            for (;;) {
                unsigned long long old = tab[slot];
                if (old == EMPTY) {
                    old = atomicCAS(&tab[slot], EMPTY, word);
                    if (old == EMPTY) break;                             // claimed for this key
                }
                if ((old >> 13) == key) { atomicMin(&tab[slot], word); break; }
                slot = slot + 1 == T ? 0u : slot + 1;
            }
            sslot[ci] = (int)slot;
        }
    }
    if (__syncthreads_or(bad)) return -1;
    SPEC_T();
    int S = 0;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        if (it * nt >= CH) break;                                        // (uniform)
        const int ci = tid + it * nt;
        int head = 0, slot = 0;
        if (ci < CH) {
            slot = sslot[ci];
            head = (int)(tab[slot] & 0x1FFFu) == ci;
        }
        int tot;
        const int ex = block_excl_scan(head, &tot, wsum);               // (also fences the reads above from the writes below)
        if (head) {
            const int run = S + ex;
            tab[slot] = (tab[slot] & ~0x1FFFull) | (unsigned)run;       // the window's row, for its other children
            if (row_base + run < capRows) {
                const size_t row = (size_t)(row_base + run);
                urois[5 * row] = 0.0f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    urois[5 * row + 1 + q] = (float)(cb[it][q] * scale);
                    ubox[4 * row + q] = cb[it][q];
                }
            }
        }
        S += tot;
    }
    __syncthreads();
    SPEC_T();
    for (int ci = tid; ci < CH; ci += nt) crow_g[ci] = (int)(tab[sslot[ci]] & 0x1FFFu);
    SPEC_T();
#ifdef AZ_SPEC_TIMING
    if (tid == 0) { printf("spec stage P=%d CH=%d S=%d (x10ns):", P, CH, S); for (int i = 1; i < tqn; ++i) printf(" %llu", tq[i] - tq[i - 1]); printf("\n"); }
#endif
    if (row_base + S > capRows) return -1;
    return S;
}
