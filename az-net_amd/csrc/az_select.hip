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
// counts, for the 256 boxes of block ib, the boxes of every NMS_RANK_JS-th 256-box block, and adds its share to rank[]
// (32 workgroups walking all of N each took 320 us of the 1.04 ms at 8129 boxes).
constexpr int NMS_RANK_JS = 8;
__global__ void __launch_bounds__(256) k_nms_rank_count(const float *__restrict__ dets, int n, int *rank)
{
    __shared__ float ss[256];
    const int nblk = (n + 255) / 256;
    const int ib = blockIdx.x, js = blockIdx.y;
    const int i = ib * 256 + threadIdx.x;
    const float si = i < n ? dets[5 * (size_t)i + 4] : 0.f;
    int cnt = 0;
    for (int t = js; t < nblk; t += NMS_RANK_JS) {
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

__global__ void __launch_bounds__(256)
k_nms_mask(const float *__restrict__ sdets, int n, double thresh, unsigned long long *mask)
{
    const int W = (n + 63) / 64;
    const int cb = blockIdx.x, rb = blockIdx.y;
    if (cb < rb) return;                       // every j in this word <= every i: nothing to suppress
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = cb * 64 + lane;
    float jx1 = 0, jy1 = 0, jx2 = 0, jy2 = 0, jarea = 0;
    if (j < n) {
        const float *d = sdets + 5 * (size_t)j;
        jx1 = d[0]; jy1 = d[1]; jx2 = d[2]; jy2 = d[3]; jarea = d[4];
    }
    for (int rr = wave; rr < 64; rr += 4) {
        const int i = rb * 64 + rr;
        if (i >= n) break;
        const float *d = sdets + 5 * (size_t)i;
        const float ix1 = d[0], iy1 = d[1], ix2 = d[2], iy2 = d[3], iarea = d[4];
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
        const bool sup = (j < n) && (j > i) && ((double)ovr >= thresh);
        const unsigned long long word = __ballot(sup);
        if (lane == 0) mask[(size_t)i * W + cb] = word;
    }
}

// One workgroup of 16 waves walks the 64-box chunks in order; the serial chain runs inside wave 0 alone.  Step c:
//   wave 0     resolves the 64 x 64 diagonal block of chunk c (its words were requested a step ahead; only boxes still
//              alive are visited: s_ff1 over the alive word), publishes the kept rows, and at once ORs THEIR words of
//              chunk c + 1 -- one load per lane, all in flight together, a butterfly -- into removed[c + 1]: all that the
//              next step's resolve still lacks;
//   the others OR the kept rows of chunk c - 1 (published a step ago) into the words c + 1 .. W - 1 -- thread = (word,
//              slot), slot takes every eighth kept row, eight neighbouring lanes meet with three shuffles;
//   one barrier.
// The chain per step is a register scan + ONE memory round trip (it was: diagonal round trip, scan, barrier, ~3 dependent
// batches of row loads on a quarter of the threads, barrier).
constexpr int NMS_SCAN_NT = 1024;
__global__ void __launch_bounds__(NMS_SCAN_NT)
k_nms_scan(const unsigned long long *__restrict__ mask, const int *__restrict__ order, int n,
           unsigned long long *removed_g, long long *keep, int *nkeep, unsigned seq)
{
    // seq != 0: keep / nkeep are host-mapped memory the host polls.  Writes of one wave to host memory may land out of
    // order (PCIe posted writes with relaxed ordering: a later word can pass an earlier one -- measured: a count visible
    // before the last keep entries, 5 calls in 27 600), so nothing is inferred from ORDER: every word carries the call's
    // sequence number in its upper half and the host takes a word only once it shows the current number.
    __shared__ int s_nkept[2];                        // kept rows of the chunk resolved in this / the previous step
    __shared__ int s_krow[2][64];                     // ... which rows (0..63), ascending
    extern __shared__ unsigned long long removed[];   // W words
    const int W = (n + 63) / 64;
    (void)removed_g;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int w = tid; w < W; w += NMS_SCAN_NT) removed[w] = 0ull;
    if (tid < 2) s_nkept[tid] = 0;
    int nk_total = 0;                                 // (wave 0) boxes kept so far
    unsigned long long diag_next = (tid < 64 && lane < n) ? mask[(size_t)lane * W] : 0ull;
    __syncthreads();
    for (int c = 0; c < W; ++c) {
        const int cur = c & 1;
        if (tid < 64) {
            const int row = c * 64 + lane;
            const unsigned long long diag = diag_next;
            const int rown = (c + 1) * 64 + lane;
            diag_next = (c + 1 < W && rown < n) ? mask[(size_t)rown * W + (c + 1)] : 0ull;
            const int nvalid = min(64, n - c * 64);
            unsigned long long alive = ~removed[c];
            if (nvalid < 64) alive &= ((1ull << nvalid) - 1ull);
            unsigned long long kept = 0ull;
            const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
            while (alive) {                           // (wave-uniform: alive comes from LDS and readlanes)
                const int b = __builtin_ctzll(alive);
                const unsigned long long drow =
                    ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)dhi, b) << 32) |
                    (unsigned)__builtin_amdgcn_readlane((int)dlo, b);
                kept |= (1ull << b);
                alive &= ~(drow | (1ull << b));
            }
            const int nkc = __popcll(kept);
            const bool mine = (kept >> lane) & 1ull;
            const int pos = __popcll(kept & ((1ull << lane) - 1ull));
            if (mine) s_krow[cur][pos] = lane;
            if (lane == 0) s_nkept[cur] = nkc;
            // the kept rows' words of the NEXT chunk: lane q takes kept row q (the row lists above are this wave's own
            // writes: krow of lane q is found with a ballot-free select below)
            if (c + 1 < W) {
                // lane `pos` of a kept lane is its index among the kept rows; invert with a permute through LDS written above
                __builtin_amdgcn_wave_barrier();
                unsigned long long v = lane < nkc ? mask[(size_t)(c * 64 + s_krow[cur][lane]) * W + (c + 1)] : 0ull;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const unsigned lo = __shfl_xor((unsigned)v, d, 64), hi = __shfl_xor((unsigned)(v >> 32), d, 64);
                    v |= ((unsigned long long)hi << 32) | lo;
                }
                if (lane == 0 && v) atomicOr(&removed[c + 1], v);      // (the helpers add chunk c - 1's share to the same word)
            }
            // append kept boxes (sorted positions -> original indices, visiting order)
            if (mine) keep[nk_total + pos] = ((long long)seq << 32) | (unsigned)order[row];
            nk_total += nkc;
        } else if (c > 0) {
            // helpers: chunk c - 1's kept rows into the words c + 1 ..: thread = (word, slot)
            const int prev = cur ^ 1, nkp = s_nkept[prev], ht = tid - 64;
            for (int w0 = c + 1; w0 < W; w0 += (NMS_SCAN_NT - 64) / 8) {
                const int w = w0 + (ht >> 3), slot = ht & 7;
                unsigned long long acc = 0ull;
                if (w < W) {
                    unsigned long long m[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int q = slot + 8 * j;
                        m[j] = q < nkp ? mask[(size_t)((c - 1) * 64 + s_krow[prev][q]) * W + w] : 0ull;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc |= m[j];
                }
#pragma unroll
                for (int d = 1; d < 8; d <<= 1) {
                    const unsigned lo = __shfl_xor((unsigned)acc, d, 64), hi = __shfl_xor((unsigned)(acc >> 32), d, 64);
                    acc |= ((unsigned long long)hi << 32) | lo;
                }
                if (w < W && slot == 0 && acc) atomicOr(&removed[w], acc);
            }
        }
        __syncthreads();
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (seq) *reinterpret_cast<long long *>(nkeep) = ((long long)seq << 32) | (unsigned)nk_total;
        else *nkeep = nk_total;
    }
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
        // (rows r >= (cb + 1) * 64 suppress nothing in this block: j > r never holds)
        const int rend = min(n, (cb + 1) * 64);
        for (int r = wave; r < rend; r += NWV) {
            const float ix1 = sd[r][0], iy1 = sd[r][1], ix2 = sd[r][2], iy2 = sd[r][3], iarea = sd[r][4];
            bool sup = false;
            if (jin && j > r) {
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
                unsigned long long kept = 0ull;
                const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
                for (int b = 0; b < nvalid; ++b) {
                    const unsigned long long drow =
                        ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)dhi, b) << 32) |
                        (unsigned)__builtin_amdgcn_readlane((int)dlo, b);
                    if ((alive >> b) & 1ull) { kept |= (1ull << b); alive &= ~drow; }
                }
                const bool mine = (kept >> lane) & 1ull;
                if (mine) keep[o + nk + __popcll(kept & ((1ull << lane) - 1ull))] = ((long long)seq << 32) | (unsigned)sorder[row];
                nk += __popcll(kept);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if (w > c && w < W) {
                        unsigned long long v = mine ? smask[row][w] : 0ull;
#pragma unroll
                        for (int dd = 32; dd > 0; dd >>= 1) {
                            const unsigned lo = __shfl_xor((unsigned)v, dd, 64), hi = __shfl_xor((unsigned)(v >> 32), dd, 64);
                            v |= ((unsigned long long)hi << 32) | lo;
                        }
                        rem[w] |= v;
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

void azk_nms(hipStream_t s, const float *dets, int n, double thresh, int *order, float *sdets,
             unsigned long long *mask, unsigned long long *removed, long long *keep, int *nkeep, unsigned seq)
{
    if (n <= 0) { hipMemsetAsync(nkeep, 0, sizeof(int), s); return; }
    const int W = (n + 63) / 64;
    const int g = (n + 255) / 256;
    // (`removed`: n ints of rank scratch, zero on entry and left zero)
    int *rank = reinterpret_cast<int *>(removed);
    hipLaunchKernelGGL(k_nms_rank_count, dim3(g, NMS_RANK_JS), dim3(256), 0, s, dets, n, rank);
    hipLaunchKernelGGL(k_nms_rank_place, dim3(g), dim3(256), 0, s, dets, n, rank, order, sdets);
    hipLaunchKernelGGL(k_nms_mask, dim3(W, W), dim3(256), 0, s, sdets, n, thresh, mask);
    hipLaunchKernelGGL(k_nms_scan, dim3(1), dim3(NMS_SCAN_NT), (size_t)W * sizeof(unsigned long long), s, mask, order,
                       n, removed, keep, nkeep, seq);
}
