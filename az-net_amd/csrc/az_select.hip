// az_select.hip -- final proposal selection (lib/detect/test.py:393-401) and greedy NMS
// (lib/utils/nms.pyx:17-68) on gfx950.  Integer / comparison work: a radix select over the
// score bits instead of a full sort, wave ballots for the NMS suppression bitmask, and a
// 64-box-at-a-time scan over that mask.  Built with -ffp-contract=off (IoU in f32 with
// one rounding per operation and IEEE division, as the Cython code computes it).
#include "az_dev.h"

namespace {

__device__ int block_excl_scan1024(int v, int *total, int *wsum)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    __syncthreads();
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

constexpr int TOPK_MAX = 4096;

// Top-k of N scores, descending, ties broken by the lower candidate index (a stable
// descending sort; NumPy's argsort(-aScores) is unstable, tied candidates are duplicates of
// one roi and carry identical boxes).  Single workgroup: 3 radix-select passes over the
// order-preserving key (11 + 11 + 10 bits) find the k-th largest key T; everything above T
// plus the first (k - count_above) items equal to T is gathered in index order, then ranked.
__global__ void __launch_bounds__(1024)
k_topk(const float *__restrict__ scores, const int *Nptr, int capN, int k, int min_n, int *sel_idx, int *nsel,
       const double *__restrict__ Yall, const float *__restrict__ Sall, double *Yout, float *Sout)
{
    __shared__ int hist[2048];
    __shared__ int wsum[17];
    __shared__ unsigned s_prefix, s_mask;
    __shared__ int s_need;
    __shared__ unsigned skey[TOPK_MAX];
    __shared__ int sidx[TOPK_MAX];
    __shared__ int s_ngt, s_neq;

    int N = *Nptr;
    if (N > capN) N = capN;
    if (min_n > 0 && N <= min_n) return;          // the counting kernels own this size
    if (k > TOPK_MAX) k = TOPK_MAX;
    const int ksel = k < N ? k : N;
    const int tid = threadIdx.x;
    unsigned T = 0;
    int need_eq = 0;
    if (ksel < N) {
        if (tid == 0) { s_prefix = 0; s_mask = 0; s_need = ksel; }
        __syncthreads();
        const int shifts[3] = {21, 10, 0};
        const int widths[3] = {11, 11, 10};
        for (int p = 0; p < 3; ++p) {
            const int sh = shifts[p], nb = 1 << widths[p];
            for (int b = tid; b < 2048; b += blockDim.x) hist[b] = 0;
            __syncthreads();
            const unsigned prefix = s_prefix, mask = s_mask;
            // eight independent (clamped, unconditional) loads in flight per thread, then the atomics
            for (int base = 0; base < N; base += 8 * (int)blockDim.x) {
                unsigned k8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) k8[j] = score_key(scores[min(base + j * (int)blockDim.x + tid, N - 1)]);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (base + j * (int)blockDim.x + tid < N && (k8[j] & mask) == prefix)
                        atomicAdd(&hist[(k8[j] >> sh) & (nb - 1)], 1);
            }
            __syncthreads();
            if (tid < 64) {
                // lane L owns bins [nb - (L+1)*per, nb - L*per): descending chunks
                const int per = nb / 64;
                const int hi = nb - tid * per;
                int csum = 0;
                for (int b = hi - 1; b >= hi - per; --b) csum += hist[b];
                int inc = csum;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    int t = __shfl_up(inc, d, 64);
                    if (tid >= d) inc += t;
                }
                const int need = s_need;
                const int before = inc - csum;           // items in chunks above this lane's
                const bool mine = (before < need) && (inc >= need);
                if (mine) {
                    int run = before, b = hi - 1;
                    for (; b >= hi - per; --b) {
                        if (run + hist[b] >= need) break;
                        run += hist[b];
                    }
                    s_prefix = prefix | ((unsigned)b << sh);
                    s_mask = mask | ((unsigned)(nb - 1) << sh);
                    s_need = need - run;                 // still needed inside bin b
                }
            }
            __syncthreads();
        }
        T = s_prefix;
        need_eq = s_need;
    }
    // ordered gather (index order) of keys > T, and of the first need_eq keys == T: thread t owns
    // the contiguous candidates [t*per, (t+1)*per), so ONE pair of block scans places everything
    const int per = (N + (int)blockDim.x - 1) / (int)blockDim.x;
    const int i0 = tid * per, i1 = min(N, i0 + per);
    int cgt = 0, ceq = 0;
    for (int b0 = i0; b0 < i1; b0 += 8) {
        unsigned k8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) k8[j] = score_key(scores[min(b0 + j, N - 1)]);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (b0 + j < i1) {
                if (ksel == N) cgt += 1;
                else { cgt += k8[j] > T; ceq += k8[j] == T; }
            }
    }
    int tot_gt, tot_eq;
    int ogt = block_excl_scan1024(cgt, &tot_gt, wsum);
    int oeq = block_excl_scan1024(ceq, &tot_eq, wsum);
    (void)s_ngt; (void)s_neq;
    for (int b0 = i0; b0 < i1; b0 += 8) {
        unsigned k8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) k8[j] = score_key(scores[min(b0 + j, N - 1)]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = b0 + j;
            if (i >= i1) continue;
            const unsigned key = k8[j];
            const bool gt = (ksel == N) || key > T;
            const bool eq = (ksel != N) && key == T;
            if (gt) {
                if (ogt < TOPK_MAX) { skey[ogt] = key; sidx[ogt] = i; }
                ++ogt;
            } else if (eq) {
                // equal keys go after all greater keys: tail slots [ksel - need_eq, ksel)
                if (oeq < need_eq) { skey[ksel - need_eq + oeq] = key; sidx[ksel - need_eq + oeq] = i; }
                ++oeq;
            }
        }
    }
    __syncthreads();
    // rank the ksel gathered items: descending key, then ascending index
    for (int a = tid; a < ksel; a += blockDim.x) {
        const unsigned ka = skey[a];
        const int ia = sidx[a];
        int rank = 0;
        for (int b = 0; b < ksel; ++b) {
            const unsigned kb = skey[b];
            rank += (kb > ka) | ((kb == ka) & (sidx[b] < ia));
        }
        sel_idx[rank] = ia;
        if (Yout) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Yout[(size_t)rank * 4 + q] = Yall[(size_t)ia * 4 + q];
            Sout[rank] = Sall[ia];
        }
    }
    if (tid == 0) *nsel = ksel;
}

// ---- top-k by counting, across the whole chip ------------------------------------------------
// rank(i) = number of candidates that sort before i (higher score, or equal score and lower
// index): the proposal at output position r is the candidate of rank r, so no sort and no
// select are needed -- only N^2 compares, which 256 CUs finish in a few microseconds for the
// N <= ~30k of this path.  Workgroup (ib, jb) counts, for its 256 candidates, the earlier-
// sorting candidates inside j-range jb (keys staged through LDS, read as broadcasts); the
// RANK_J partial counts of a candidate are summed by the scatter kernel.  Integer work:
// identical to the radix-select kernel above, bit for bit.
constexpr int RANK_J = 32;
constexpr int RANK_TILE = 1024;
constexpr int RANK_MAX_N = 65536;      // beyond this the single-workgroup radix select takes over

__device__ __forceinline__ unsigned long long rank_comp(float score, int idx)
{
    return ((unsigned long long)score_key(score) << 32) | (unsigned)(~(unsigned)idx);
}

__global__ void __launch_bounds__(256)
k_rank_count(const float *__restrict__ scores, const int *Nptr, int capN, int *__restrict__ part)
{
    __shared__ __attribute__((aligned(16))) unsigned long long sk[RANK_TILE];
    int N = *Nptr;
    if (N > capN) N = capN;
    if (N > RANK_MAX_N) return;
    const int i0 = blockIdx.x * 256;
    if (i0 >= N) return;
    const int i = i0 + threadIdx.x;
    const unsigned long long ci = i < N ? rank_comp(scores[i], i) : ~0ull;
    const int jlen = (N + RANK_J - 1) / RANK_J;
    const int jb = blockIdx.y * jlen, je = min(N, jb + jlen);
    int cnt = 0;
    for (int t0 = jb; t0 < je; t0 += RANK_TILE) {
        __syncthreads();
        for (int q = threadIdx.x; q < RANK_TILE; q += 256) {
            const int j = t0 + q;
            sk[q] = j < je ? rank_comp(scores[j], j) : 0ull;        // 0 sorts after everything
        }
        __syncthreads();
        const int lim = min(RANK_TILE, je - t0);
        const int lim4 = (lim + 3) & ~3;                             // padding entries are 0: never counted
#pragma unroll 4
        for (int q = 0; q < lim4; q += 2) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(&sk[q]);
            cnt += (v.x > ci) ? 1 : 0;
            cnt += (v.y > ci) ? 1 : 0;
        }
    }
    if (i < N) part[(size_t)blockIdx.y * capN + i] = cnt;
}

__global__ void __launch_bounds__(256)
k_rank_scatter(const float *__restrict__ scores, const int *Nptr, int capN, int k, const int *__restrict__ part,
               int *sel_idx, int *nsel, const double *__restrict__ Yall, const float *__restrict__ Sall,
               double *Yout, float *Sout)
{
    int N = *Nptr;
    if (N > capN) N = capN;
    if (N > RANK_MAX_N) return;
    if (k > TOPK_MAX) k = TOPK_MAX;
    const int ksel = k < N ? k : N;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *nsel = ksel;
    if (i >= N) return;
    int r = 0;
#pragma unroll
    for (int j = 0; j < RANK_J; ++j) r += part[(size_t)j * capN + i];
    if (r < ksel) {
        sel_idx[r] = i;
        if (Yout) {
#pragma unroll
            for (int q = 0; q < 4; ++q) Yout[(size_t)r * 4 + q] = Yall[(size_t)i * 4 + q];
            Sout[r] = Sall[i];
        }
    }
}

// aScores >= Tc selection (lib/detect/test.py:393-395), original order, double compare.
__global__ void __launch_bounds__(1024)
k_thresh_select(const float *__restrict__ scores, const int *Nptr, int capN, double Tc, int cap_out,
                int *sel_idx, int *nsel, const double *__restrict__ Yall, const float *__restrict__ Sall,
                double *Yout, float *Sout)
{
    __shared__ int wsum[17];
    int N = *Nptr;
    if (N > capN) N = capN;
    int run = 0;
    for (int base = 0; base < N; base += blockDim.x) {
        const int i = base + threadIdx.x;
        const int fl = (i < N) && ((double)scores[i] >= Tc);
        int tot;
        const int off = block_excl_scan1024(fl, &tot, wsum);
        const int dst = run + off;
        if (fl && dst < cap_out) {
            sel_idx[dst] = i;
            if (Yout) {
#pragma unroll
                for (int q = 0; q < 4; ++q) Yout[(size_t)dst * 4 + q] = Yall[(size_t)i * 4 + q];
                Sout[dst] = Sall[i];
            }
        }
        run += tot;
    }
    if (threadIdx.x == 0) *nsel = run;      // may exceed cap_out: the host reports AZ_ERR_CAPACITY
}

// ======================================================================================
// NMS, lib/utils/nms.pyx:17-68.
// 1. order = argsort(scores)[::-1] as a rank sort (descending score; equal scores: higher
//    original index first, which is what a stable ascending sort reversed yields).
// 2. suppression bitmask: mask[i][w] bit b = (j = 64w + b > i) && IoU(i, j) >= thresh, built
//    with one wave ballot per (row, 64-column word).
// 3. greedy scan 64 sorted boxes at a time: the 64x64 diagonal word block resolves the
//    dependencies inside the chunk, then the kept rows are OR-ed into the removed bitmap.
// ======================================================================================
// rank(i) = #{j: s_j > s_i or (s_j == s_i and j > i)}: the N^2 compares are cut over the whole chip -- workgroup (ib, js)
// counts, for the 256 boxes of block ib, the boxes of every gridDim.y-th 256-box block, and adds its share to rank[]
// (32 workgroups walking all of N each took 320 us of the 1.04 ms at 8129 boxes).
// (32 column shares -- four waves per SIMD at 8129 boxes -- instead of 8: 33.0 -> 14.5 us)
constexpr int NMS_RANK_JS = 32;
__global__ void __launch_bounds__(256) k_nms_rank_count(const float *__restrict__ dets, int n, int *rank)
{
    __shared__ float ss[256];
    const int nblk = (n + 255) / 256;
    const int ib = blockIdx.x, js = blockIdx.y;
    const int i = ib * 256 + threadIdx.x;
    const float si = i < n ? dets[5 * (size_t)i + 4] : 0.f;
    int cnt = 0;
    for (int t = js; t < nblk; t += (int)gridDim.y) {
        const int j = t * 256 + threadIdx.x;
        __syncthreads();
        ss[threadIdx.x] = j < n ? dets[5 * (size_t)j + 4] : 0.f;
        __syncthreads();
        const int lim = min(256, n - t * 256);
#pragma unroll 8
        for (int jj = 0; jj < lim; ++jj) {
            const float sj = ss[jj];
            const int jidx = t * 256 + jj;
            cnt += (sj > si) | ((sj == si) & (jidx > i));
        }
    }
    if (i < n && cnt) atomicAdd(&rank[i], cnt);
}

__global__ void __launch_bounds__(256) k_nms_rank_place(const float *__restrict__ dets, int n, int *rank, int *order, float *sdets)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = rank[i];
    rank[i] = 0;                                   // (left clean for the next call)
    order[r] = i;
    const float x1 = dets[5 * (size_t)i], y1 = dets[5 * (size_t)i + 1];
    const float x2 = dets[5 * (size_t)i + 2], y2 = dets[5 * (size_t)i + 3];
    float w = x2 - x1; w = w + 1.0f;
    float h = y2 - y1; h = h + 1.0f;
    float *o = sdets + 5 * (size_t)r;
    o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = w * h;      // areas, nms.pyx:24
}

// OR over the 64 lanes (wave-uniform result): data-parallel-primitive moves in the vector ALU -- a scan inside each row of
// 16 lanes, then the rows' last lanes handed on (row_bcast) -- instead of 12 LDS permutes per 64-bit word
__device__ __forceinline__ unsigned wave_or32(unsigned x)
{
    int v = (int)x;
    v |= __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v |= __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v |= __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v |= __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v |= __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
    v |= __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
    return (unsigned)__builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ unsigned long long or_reduce(unsigned long long v)
{
    return ((unsigned long long)wave_or32((unsigned)(v >> 32)) << 32) | wave_or32((unsigned)v);
}

__global__ void __launch_bounds__(256)
k_nms_mask(const float *__restrict__ sdets, int n, double thresh, unsigned long long *mask, unsigned long long *band,
           int near)
{
    const int W = (n + 63) / 64;
    const int cb = blockIdx.x, rb = blockIdx.y;
    if (cb < rb) return;                       // every j in this word <= every i: nothing to suppress
    const int lane = threadIdx.x & 63;
    // (the wave number in a scalar register: the row's box then comes in with scalar loads, requested rows ahead)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = cb * 64 + lane;
    float jx1 = 0, jy1 = 0, jx2 = 0, jy2 = 0, jarea = 0;
    if (j < n) {
        const float *d = sdets + 5 * (size_t)j;
        jx1 = d[0]; jy1 = d[1]; jx2 = d[2]; jy2 = d[3]; jarea = d[4];
    }
    // the wave's 16 rows, four at a time: their boxes requested together (scalar loads: the addresses are wave-uniform)
    const int nrow = min(64, n - rb * 64);
    for (int q0 = 0; q0 < 16; q0 += 4) {
        float bx[4][5];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float *d = sdets + 5 * (size_t)(rb * 64 + min(wave + 4 * (q0 + u), nrow - 1));
#pragma unroll
            for (int f = 0; f < 5; ++f) bx[u][f] = d[f];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = wave + 4 * (q0 + u);
            if (rr >= nrow) break;
            const int i = rb * 64 + rr;
            const float ix1 = bx[u][0], iy1 = bx[u][1], ix2 = bx[u][2], iy2 = bx[u][3], iarea = bx[u][4];
            const float xx1 = ix1 >= jx1 ? ix1 : jx1;          // nms.pyx:11-15 max/min
            const float yy1 = iy1 >= jy1 ? iy1 : jy1;
            const float xx2 = ix2 <= jx2 ? ix2 : jx2;
            const float yy2 = iy2 <= jy2 ? iy2 : jy2;
            float tw = xx2 - xx1; tw = tw + 1.0f;
            float th = yy2 - yy1; th = th + 1.0f;
            const float w = 0.0f >= tw ? 0.0f : tw;
            const float h = 0.0f >= th ? 0.0f : th;
            const float inter = w * h;
            float den = iarea + jarea; den = den - inter;
            const float ovr = inter / den;
            // (the DIAGONAL block is written symmetric -- bit j also for j < i: the scan resolves a chunk from every box's
            //  EARLIER overlapping boxes, which is this word below its own bit; IoU is symmetric bit for bit: max / min / + / *
            //  commute.  Off the diagonal only j > i exists.)
            const bool sup = (j < n) && (cb == rb ? j != i : j > i) && ((double)ovr >= thresh);
            const unsigned long long word = __ballot(sup);
            if (lane == 0) {
                mask[(size_t)i * W + cb] = word;
                // the chunk's diagonal block and the `near` blocks right of it once more, block by block (64 row words side
                // by side): what the scan's chain needs per chunk comes in with coalesced 512-byte loads
                if (cb - rb <= near) band[((size_t)rb * (1 + near) + (cb - rb)) * 64 + rr] = word;
            }
        }
    }
}

// The greedy scan: one workgroup of 16 waves, NO barrier in its loop.  mask[i][w] = the suppression word of sorted box i
// against the 64 boxes of chunk w (row-major: a row's words lie side by side); band[c][q][r] = the same words of chunk c's
// rows for the chunks c .. c + NMS_NEAR, block by block (k_nms_mask writes both).
//   wave 0 (the serial chain): for chunk c = 0, 1, ...: waits until the helpers have applied chunks 0 .. c - NMS_NEAR - 1 (a
//     progress word per helper in LDS), resolves the chunk in ROUNDS from every box's earlier overlapping boxes (the
//     diagonal block is symmetric), publishes kept[c] in LDS, and ORs the kept rows' words of chunks c + 1 .. c + NMS_NEAR
//     into removed[] itself.  It issues no load from memory: its blocks come out of an LDS ring that
//   wave 1 (the loader) fills NMS_LB chunks at a time from `band`, up to NMS_RING chunks ahead of the chain.
//   waves 2-13 (helpers, NMS_TEAMS teams of NMS_TEAM): LANE = WORD.  Team t takes the chunks t, t + NMS_TEAMS, ...; its
//     members share the chunk's rows; once kept[j] is published a member fetches the words of ITS KEPT rows (64 words per
//     load, far words only: j + NMS_NEAR + 1 on), ORs them in-lane and ORs the lanes' words into removed[] with ONE LDS
//     atomic per 64 words.  The loads depend on kept[j], so a chunk costs its team one memory round trip -- which it has:
//     the words are needed NMS_NEAR + 1 steps after the publication, and the other teams take the chunks in between.
//     (Earlier forms fetched ALL rows ahead of the publication, 72 KB per chunk through ONE CU's L1: that, not the chain,
//     set the pace -- 1.5 us per chunk; a wave per 64 x 64 block with an OR ACROSS lanes was bound by one CU's vector ALUs.)
// No wave waits for something that waits for it: wave 0 at chunk c needs chunks <= c - NMS_NEAR - 1 applied, whose teams
// need only kept[j], published NMS_NEAR + 1 or more steps ago, and ring data the loader fetches as soon as chunk
// c - NMS_RING is done; every spin is on LDS, with s_sleep.
constexpr int NMS_SCAN_NT = 1024;
#ifndef AZ_NMS_NEAR
#define AZ_NMS_NEAR 4
#endif
#ifndef AZ_NMS_SLEEP
#define AZ_NMS_SLEEP 1
#endif
#ifndef AZ_NMS_TEAMS
#define AZ_NMS_TEAMS 3
#endif
constexpr int NMS_NEAR = AZ_NMS_NEAR;             // words ahead that wave 0 serves itself
constexpr int NMS_RING = 16, NMS_LB = 8;          // chunks in the LDS ring; chunks the loader fetches per round trip
constexpr int NMS_SLOT_WORDS = 64 * (1 + NMS_NEAR);                   // u64 per ring slot (+ 64 ints of `order`)
constexpr int NMS_TEAMS = AZ_NMS_TEAMS, NMS_TEAM = 12 / NMS_TEAMS;   // helper teams; helpers per team
constexpr int NMS_NH = NMS_TEAMS * NMS_TEAM;      // helper waves (waves 2 .. NMS_NH + 1)
static_assert(NMS_NH <= NMS_SCAN_NT / 64 - 2 && NMS_TEAM >= 1, "the helpers are waves 2 .. 15 at most");
__global__ void __launch_bounds__(NMS_SCAN_NT)
k_nms_scan(const unsigned long long *__restrict__ mask, const unsigned long long *__restrict__ band,
           const int *__restrict__ order, int n, unsigned long long *removed_g, long long *keep, int *nkeep, unsigned seq)
{
    // seq != 0: keep / nkeep are host-mapped memory the host polls.  Writes of one wave to host memory may land out of
    // order (PCIe posted writes with relaxed ordering: a later word can pass an earlier one -- measured: a count visible
    // before the last keep entries, 5 calls in 27 600), so nothing is inferred from ORDER: every word carries the call's
    // sequence number in its upper half and the host takes a word only once it shows the current number.
    extern __shared__ unsigned long long nms_lds[];
    const int W = (n + 63) / 64;
    unsigned long long *ring = nms_lds;                                       // [NMS_RING][1 + NMS_NEAR][64]
    int *ring_o = reinterpret_cast<int *>(ring + NMS_RING * NMS_SLOT_WORDS);  // [NMS_RING][64]
    unsigned long long *removed = reinterpret_cast<unsigned long long *>(ring_o + NMS_RING * 64);       // [W]
    unsigned long long *kept_w = removed + W;                 // [W] kept rows of chunk c (valid once pub > c)
    int *hdone = reinterpret_cast<int *>(kept_w + W);         // [64] the next chunk helper h takes (lanes >= NMS_NH: "all")
    int *pub = hdone + 64;                                    // chunks wave 0 has published
    int *loaded = pub + 1;                                    // chunks whose blocks are in the ring
    (void)removed_g;
    // (LDS words that another wave writes are read and written as relaxed workgroup-scope atomics -- ds_read / ds_write; a
    //  `volatile` access through a generic pointer became a system-scope FLAT instruction with a vmcnt(0) wait behind it)
    auto ld32 = [](int *q) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); };
    auto ld32a = [](int *q) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(q, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)); };
    auto ld64u = [](unsigned long long *q) {                 // wave-uniform result in scalar registers
        const unsigned long long v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32) |
               (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int w = tid; w < W; w += NMS_SCAN_NT) removed[w] = 0ull;
    if (tid < 64) hdone[tid] = tid < NMS_NH ? tid / NMS_TEAM : 0x7fffffff;    // (a helper's first chunk = its team's number)
    if (tid == 0) { *pub = 0; *loaded = 0; }
    __syncthreads();
    if (wave == 0) {
        int nk_total = 0;
        for (int c = 0; c < W; ++c) {
            const int slot = c & (NMS_RING - 1);
            if ((c & (NMS_LB - 1)) == 0)
                while (ld32a(loaded) <= c) __builtin_amdgcn_s_sleep(AZ_NMS_SLEEP);
            const unsigned long long dg = ring[slot * NMS_SLOT_WORDS + lane];
            unsigned long long ah[NMS_NEAR];
#pragma unroll
            for (int q = 0; q < NMS_NEAR; ++q) ah[q] = ring[slot * NMS_SLOT_WORDS + (1 + q) * 64 + lane];
            const int og = ring_o[slot * 64 + lane];
            const int need = c - NMS_NEAR > 0 ? c - NMS_NEAR : 0;
            // every helper is past the chunks 0 .. need - 1 of its team (lane h reads helper h's word; the others read "all")
            while (__ballot(__hip_atomic_load(&hdone[lane], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= need) != ~0ull)
                __builtin_amdgcn_s_sleep(AZ_NMS_SLEEP);
            const int nvalid = min(64, n - c * 64);
            unsigned long long alive = ~ld64u(&removed[c]);
            if (nvalid < 64) alive &= ((1ull << nvalid) - 1ull);
            // Greedy inside the chunk, in rounds instead of box by box: a box still undecided whose earlier overlapping boxes
            // are all decided-and-dropped is kept (the sequential walk would keep it: nothing kept lies before it that
            // suppresses it); a box with a kept earlier overlapping box is dropped.  Every round decides at least the first
            // undecided box; the rounds needed are the longest chain of overlaps, a handful -- against one dependent
            // s_ff1 / v_readlane step per KEPT box (~24 per chunk at 8129 boxes).  Same kept set: the greedy one is unique.
            unsigned long long kept = 0ull, und = alive;                         // (wave-uniform: scalar registers)
            const unsigned long long pred = dg & ((1ull << lane) - 1ull);        // earlier boxes of the chunk that overlap this one
            while (und) {
                const bool in_und = (und >> lane) & 1ull;
                const unsigned long long newk = __ballot(in_und && (pred & (und | kept)) == 0ull);
                kept |= newk;
                const unsigned long long gone = __ballot(in_und && (pred & kept) != 0ull);
                und &= ~(newk | gone);
            }
            const bool mine = (kept >> lane) & 1ull;
            // its own near words first (the next steps' removed words), then the publication the helpers wait for
#pragma unroll
            for (int q = 0; q < NMS_NEAR; ++q)
                if (c + 1 + q < W) {
                    // (an OR across the lanes, then ONE LDS atomic; every kept lane OR-ing its own word into LDS -- the unit
                    //  serialises a wave's atomics on one address -- measured the same: 126 against 123 us)
                    const unsigned long long v = or_reduce(mine ? ah[q] : 0ull);
                    if (lane == 0 && v) atomicOr(&removed[c + 1 + q], v);
                }
            if (lane == 0) {
                __hip_atomic_store(&kept_w[c], kept, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(pub, c + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // append kept boxes (sorted positions -> original indices, visiting order)
            const int pos = __popcll(kept & ((1ull << lane) - 1ull));
            if (mine) keep[nk_total + pos] = ((long long)seq << 32) | (unsigned)og;
            nk_total += __popcll(kept);
        }
        __threadfence_system();
        if (lane == 0) {
            if (seq) *reinterpret_cast<long long *>(nkeep) = ((long long)seq << 32) | (unsigned)nk_total;
            else *nkeep = nk_total;
        }
        return;
    }
    const size_t last_row = (size_t)(n - 1);
    if (wave == 1) {
        // the loader: chunks c0 .. c0 + NMS_LB - 1 with one round trip of coalesced loads (lane = row of the block; chunk
        // indices clamped, every load unconditional), into ring slots that wave 0 has left behind
        for (int c0 = 0; c0 < W; c0 += NMS_LB) {
            while (ld32(pub) < c0 + NMS_LB - NMS_RING) __builtin_amdgcn_s_sleep(2);
            unsigned long long blk[NMS_LB][1 + NMS_NEAR];
            int og[NMS_LB];
#pragma unroll
            for (int u = 0; u < NMS_LB; ++u) {
                const int cc = min(c0 + u, W - 1);
#pragma unroll
                for (int q = 0; q <= NMS_NEAR; ++q) blk[u][q] = band[((size_t)cc * (1 + NMS_NEAR) + q) * 64 + lane];
                og[u] = order[min((size_t)cc * 64 + lane, last_row)];
            }
#pragma unroll
            for (int u = 0; u < NMS_LB; ++u) {
                const int slot = (c0 + u) & (NMS_RING - 1);
#pragma unroll
                for (int q = 0; q <= NMS_NEAR; ++q) ring[slot * NMS_SLOT_WORDS + q * 64 + lane] = blk[u][q];
                ring_o[slot * 64 + lane] = og[u];
            }
            if (lane == 0) __hip_atomic_store(loaded, min(c0 + NMS_LB, W), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    if (wave >= 2 + NMS_NH) return;
    // helpers: team = h / NMS_TEAM takes the chunks team, team + NMS_TEAMS, ...; member k = h % NMS_TEAM the rows k, k +
    // NMS_TEAM, ... of each; GB groups of 64 words per pass over its kept rows
    constexpr int HR = (64 + NMS_TEAM - 1) / NMS_TEAM, GB = 2;
    const int h = wave - 2, team = h / NMS_TEAM, k = h - team * NMS_TEAM;
    const int G = (W + 63) / 64;
    for (int j = team; j < W; j += NMS_TEAMS) {
        const int w_lo = j + 1 + NMS_NEAR;                   // the first word that is not wave 0's own
        if (w_lo < W) {
            while (ld32a(pub) <= j) __builtin_amdgcn_s_sleep(AZ_NMS_SLEEP);
            const unsigned long long kept = ld64u(&kept_w[j]);
            for (int g0 = w_lo / 64; g0 < G; g0 += GB) {
                // (all the loads first -- one round trip --, the ORs behind them)
                unsigned long long v[HR][GB], acc[GB];
#pragma unroll
                for (int r = 0; r < HR; ++r) {
                    const int rr = k + r * NMS_TEAM;
#pragma unroll
                    for (int q = 0; q < GB; ++q) v[r][q] = 0ull;
                    if (rr < 64 && ((kept >> rr) & 1ull)) {  // (wave-uniform)
                        const unsigned long long *row = mask + ((size_t)j * 64 + rr) * W;
#pragma unroll
                        for (int q = 0; q < GB; ++q) v[r][q] = row[min((g0 + q) * 64 + lane, W - 1)];
                    }
                }
#pragma unroll
                for (int q = 0; q < GB; ++q) acc[q] = 0ull;
#pragma unroll
                for (int r = 0; r < HR; ++r)
#pragma unroll
                    for (int q = 0; q < GB; ++q) acc[q] |= v[r][q];
#pragma unroll
                for (int q = 0; q < GB; ++q) {
                    const int w = (g0 + q) * 64 + lane;
                    if (w >= w_lo && w < W && acc[q]) atomicOr(&removed[w], acc[q]);
                }
            }
        }
        if (lane == 0) __hip_atomic_store(&hdone[h], j + NMS_TEAMS, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (lane == 0) __hip_atomic_store(&hdone[h], 0x7fffffff, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- many small NMS problems in one launch (apply_nms: one per class per image) ---------------
// One workgroup per group of n <= NMS_SMALL boxes; rank sort, suppression bit matrix and greedy scan
// all in LDS.  Same arithmetic, same order conventions as the three kernels above.
constexpr int NMS_SMALL = 256;

constexpr int NMS_SMALL_NT = 1024;       // 16 waves: the suppression words are n^2 / 64 (row, 64-column block) pairs of
                                        // ~100 dependent cycles each -- 4 waves took 49 us of a 60 us kernel at 256 boxes
__global__ void __launch_bounds__(NMS_SMALL_NT)
k_nms_small(const float *__restrict__ dets, const int *__restrict__ goff, const int *__restrict__ gsel,
            int n_single, double thresh, long long *__restrict__ keep, int *__restrict__ nkeep,
            int *done_cnt, int *done_flag, int done_seq, unsigned seq)
{
    // seq != 0 (results in host-mapped memory, polled by the host): every keep entry carries seq in its upper half and the
    // group's count is (seq << 9) | n_kept -- see k_nms_scan: the host trusts tags, never the order in which words land
    __shared__ float sd[NMS_SMALL][5];                     // sorted: x1, y1, x2, y2, area
    __shared__ int sorder[NMS_SMALL];
    __shared__ float ss[NMS_SMALL];
    __shared__ unsigned long long smask[NMS_SMALL][NMS_SMALL / 64];
    // (goff == NULL: ONE problem of n_single boxes at dets, results at keep[0..], nkeep[0] -- az_nms's small case,
    //  where dets / keep / nkeep may be host-mapped memory: each is touched once)
    const int g = goff ? gsel[blockIdx.x] : 0;
    const int o = goff ? goff[g] : 0, n = goff ? goff[g + 1] - o : n_single;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the boxes come in with ONE round trip of coalesced loads (in az_nms's small case they sit in host-mapped memory: a
    // strided read per field would be a PCIe transaction per lane and field -- 23 us of kernel for 100 boxes)
    __shared__ float sraw[NMS_SMALL * 5];
    {
        const float *g = dets + 5 * (size_t)o;
        for (int j = tid; j < 5 * n; j += NMS_SMALL_NT) sraw[j] = g[j];
    }
    __syncthreads();
    const float *d = sraw;
    const int i = tid;
    const float si = i < n ? d[5 * i + 4] : 0.f;
    if (tid < NMS_SMALL) ss[tid] = si;
    __syncthreads();
    if (i < n) {
        int rank = 0;
#pragma unroll 8
        for (int j = 0; j < n; ++j) {                   // (eight independent LDS broadcasts in flight)
            const float sj = ss[j];
            rank += (sj > si) | ((sj == si) & (j > i));
        }
        const float x1 = d[5 * i], y1 = d[5 * i + 1], x2 = d[5 * i + 2], y2 = d[5 * i + 3];
        float w = x2 - x1; w = w + 1.0f;
        float h = y2 - y1; h = h + 1.0f;
        sorder[rank] = i;
        sd[rank][0] = x1; sd[rank][1] = y1; sd[rank][2] = x2; sd[rank][3] = y2; sd[rank][4] = w * h;
    }
    __syncthreads();
    const int W = (n + 63) >> 6;
    // suppression words: lane = column j of a 64-column block (its box in registers), the workgroup's 16 waves share the
    // rows (four rows per wave and turn written out by hand measured no better)
    constexpr int NWV = NMS_SMALL_NT / 64;
    for (int cb = 0; cb < W; ++cb) {
        const int j = cb * 64 + lane;
        const bool jin = j < n;
        const float jx1 = jin ? sd[j][0] : 0.f, jy1 = jin ? sd[j][1] : 0.f, jx2 = jin ? sd[j][2] : 0.f,
                    jy2 = jin ? sd[j][3] : 0.f, jarea = jin ? sd[j][4] : 0.f;
        // (rows r >= (cb + 1) * 64 suppress nothing in this block: j > r never holds, and it is not their own chunk)
        const int rend = min(n, (cb + 1) * 64);
        for (int r = wave; r < rend; r += NWV) {
            const float ix1 = sd[r][0], iy1 = sd[r][1], ix2 = sd[r][2], iy2 = sd[r][3], iarea = sd[r][4];
            bool sup = false;
            // (the block of the row's own chunk is symmetric -- bit j also for j < r --: the scan below resolves a chunk from
            //  every box's EARLIER overlapping boxes)
            if (jin && ((r >> 6) == cb ? j != r : j > r)) {
                const float xx1 = ix1 >= jx1 ? ix1 : jx1;
                const float yy1 = iy1 >= jy1 ? iy1 : jy1;
                const float xx2 = ix2 <= jx2 ? ix2 : jx2;
                const float yy2 = iy2 <= jy2 ? iy2 : jy2;
                float tw = xx2 - xx1; tw = tw + 1.0f;
                float th = yy2 - yy1; th = th + 1.0f;
                const float w = 0.0f >= tw ? 0.0f : tw;
                const float h = 0.0f >= th ? 0.0f : th;
                const float inter = w * h;
                float den = iarea + jarea; den = den - inter;
                const float ovr = inter / den;
                sup = ((double)ovr >= thresh);
            }
            const unsigned long long word = __ballot(sup);
            if (lane == 0) smask[r][cb] = word;
        }
        // (rows past this block's end: no column of the block lies behind them)
        for (int r = rend + tid; r < n; r += NMS_SMALL_NT) smask[r][cb] = 0ull;
    }
    __syncthreads();
    if (wave == 0) {
        // greedy scan over <= 256 sorted boxes by ONE wave, 64 boxes at a time (as k_nms_scan): lane = box of the block,
        // its diagonal word in registers, the 64 sequential decisions through readlane -- no memory access in the
        // dependent chain (a single thread walking smask paid two LDS latencies per box: ~10 us for 100 boxes); then
        // the block's kept rows are OR-ed into the later blocks' removed words with a butterfly.
        static_assert(NMS_SMALL / 64 == 4, "the scan below is written for four 64-box blocks");
        unsigned long long rem[4] = {0ull, 0ull, 0ull, 0ull};          // wave-uniform
        int nk = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < W) {
                const int row = c * 64 + lane;
                const unsigned long long diag = row < n ? smask[row][c] : 0ull;
                const int nvalid = min(64, n - c * 64);
                unsigned long long alive = ~rem[c];
                if (nvalid < 64) alive &= ((1ull << nvalid) - 1ull);
                // in rounds (as k_nms_scan): an undecided box none of whose earlier overlapping boxes is undecided or kept is
                // kept, one with a kept earlier overlapping box is dropped -- the greedy set, a few rounds instead of 64 steps
                unsigned long long kept = 0ull, und = alive;
                const unsigned long long pred = diag & ((1ull << lane) - 1ull);
                while (und) {
                    const bool in_und = (und >> lane) & 1ull;
                    const unsigned long long newk = __ballot(in_und && (pred & (und | kept)) == 0ull);
                    kept |= newk;
                    const unsigned long long gone = __ballot(in_und && (pred & kept) != 0ull);
                    und &= ~(newk | gone);
                }
                const bool mine = (kept >> lane) & 1ull;
                if (mine) keep[o + nk + __popcll(kept & ((1ull << lane) - 1ull))] = ((long long)seq << 32) | (unsigned)sorder[row];
                nk += __popcll(kept);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if (w > c && w < W) {
                        rem[w] |= or_reduce(mine ? smask[row][w] : 0ull);
                    }
                }
            }
        }
        __threadfence_system();
        if (lane == 0) nkeep[g] = seq ? (int)(((seq & 0x3FFFFFu) << 9) | (unsigned)nk) : nk;
        // (batched form on host-mapped memory: the LAST workgroup to finish raises the flag the host polls)
        if (done_flag && lane == 0) {
            __threadfence_system();
            if (atomicAdd(done_cnt, 1) == (int)gridDim.x - 1) {
                *done_cnt = 0;
                __threadfence_system();
                *(volatile int *)done_flag = done_seq;
            }
        }
    }
}

__global__ void k_gather_sel(const int *__restrict__ sel_idx, const int *nsel, int cap,
                             const double *__restrict__ Yall, const float *__restrict__ Sall, double *Yout,
                             float *Sout)
{
    int n = *nsel;
    if (n > cap) n = cap;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int src = sel_idx[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) Yout[(size_t)i * 4 + q] = Yall[(size_t)src * 4 + q];
        Sout[i] = Sall[src];
    }
}

}  // namespace

// --------------------------------------------------------------------------------------
int azk_topk_scratch_ints(int capN) { return RANK_J * capN; }

void azk_topk_full(hipStream_t s, const float *scores, const int *Nptr, int capN, int k, int *sel_idx,
                   int *nsel, const double *Yall, const float *Sall, double *Yout, float *Sout, int *rank_scratch)
{
    if (rank_scratch) {
        const int nb = (min(capN, RANK_MAX_N) + 255) / 256;
        hipLaunchKernelGGL(k_rank_count, dim3(nb, RANK_J), dim3(256), 0, s, scores, Nptr, capN, rank_scratch);
        hipLaunchKernelGGL(k_rank_scatter, dim3(nb), dim3(256), 0, s, scores, Nptr, capN, k, rank_scratch, sel_idx,
                           nsel, Yall, Sall, Yout, Sout);
        if (capN <= RANK_MAX_N) return;           // N can never exceed what the counting kernels take
    }
    hipLaunchKernelGGL(k_topk, dim3(1), dim3(1024), 0, s, scores, Nptr, capN, k, rank_scratch ? RANK_MAX_N : 0,
                       sel_idx, nsel, Yall, Sall, Yout, Sout);
}

void azk_topk(hipStream_t s, const float *scores, const int *Nptr, int capN, int k, int *sel_idx, int *nsel,
              int *rank_scratch)
{
    azk_topk_full(s, scores, Nptr, capN, k, sel_idx, nsel, nullptr, nullptr, nullptr, nullptr, rank_scratch);
}

void azk_thresh_select_full(hipStream_t s, const float *scores, const int *Nptr, int capN, double Tc,
                            int cap_out, int *sel_idx, int *nsel, const double *Yall, const float *Sall,
                            double *Yout, float *Sout)
{
    hipLaunchKernelGGL(k_thresh_select, dim3(1), dim3(1024), 0, s, scores, Nptr, capN, Tc, cap_out, sel_idx,
                       nsel, Yall, Sall, Yout, Sout);
}

void azk_gather_sel(hipStream_t s, const int *sel_idx, const int *nsel, int cap, const double *Yall,
                    const float *Sall, double *Yout, float *Sout)
{
    hipLaunchKernelGGL(k_gather_sel, dim3(64), dim3(256), 0, s, sel_idx, nsel, cap, Yall, Sall, Yout, Sout);
}

int azk_nms_small_max() { return NMS_SMALL; }

void azk_nms_small(hipStream_t s, const float *dets, const int *goff, const int *gsel, int n_sel, double thresh,
                   long long *keep, int *nkeep, int *done_cnt, int *done_flag, int done_seq, unsigned seq)
{
    if (n_sel > 0)
        hipLaunchKernelGGL(k_nms_small, dim3(n_sel), dim3(NMS_SMALL_NT), 0, s, dets, goff, gsel, 0, thresh, keep, nkeep,
                           done_cnt, done_flag, done_seq, seq);
}

void azk_nms_one_small(hipStream_t s, const float *dets, int n, double thresh, long long *keep, int *nkeep, unsigned seq)
{
    hipLaunchKernelGGL(k_nms_small, dim3(1), dim3(NMS_SMALL_NT), 0, s, dets, (const int *)nullptr, (const int *)nullptr, n, thresh,
                       keep, nkeep, (int *)nullptr, (int *)nullptr, 0, seq);
}

// LDS of the scan: the block ring, removed[W], kept[W] (8 bytes each), the helpers' progress words + two counters
size_t azk_nms_scan_lds_bytes(int n)
{
    const size_t W = (size_t)(n + 63) / 64;
    return (size_t)NMS_RING * (NMS_SLOT_WORDS * 8 + 64 * 4) + W * 16 + 66 * 4 + 16;
}

// words of the band array (the diagonal + near blocks of every chunk, block by block) for up to n boxes
size_t azk_nms_band_words(int n) { return ((size_t)(n + 63) / 64) * (1 + NMS_NEAR) * 64; }

void azk_nms(hipStream_t s, const float *dets, int n, double thresh, int *order, float *sdets,
             unsigned long long *mask, unsigned long long *band, unsigned long long *removed, long long *keep, int *nkeep,
             unsigned seq)
{
    if (n <= 0) { hipMemsetAsync(nkeep, 0, sizeof(int), s); return; }
    const int W = (n + 63) / 64;
    const int g = (n + 255) / 256;
    // (`removed`: n ints of rank scratch, zero on entry and left zero)
    int *rank = reinterpret_cast<int *>(removed);
    hipLaunchKernelGGL(k_nms_rank_count, dim3(g, g < NMS_RANK_JS ? g : NMS_RANK_JS), dim3(256), 0, s, dets, n, rank);
    hipLaunchKernelGGL(k_nms_rank_place, dim3(g), dim3(256), 0, s, dets, n, rank, order, sdets);
    hipLaunchKernelGGL(k_nms_mask, dim3(W, W), dim3(256), 0, s, sdets, n, thresh, mask, band, NMS_NEAR);
    // (more than the default 64 KB of dynamic LDS from ~100 000 boxes on: opted in once per process and device)
    if (azk_nms_scan_lds_bytes(n) > 60000) {
        static bool opted[64] = {false};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 0 && dev < 64 && !opted[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_nms_scan), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
            opted[dev] = true;
        }
    }
    hipLaunchKernelGGL(k_nms_scan, dim3(1), dim3(NMS_SCAN_NT), azk_nms_scan_lds_bytes(n), s, mask, band, order,
                       n, removed, keep, nkeep, seq);
}
