// az_geom.hip -- region geometry of the AZ search on gfx950: roi projection + 1/16 dedup,
// candidate filter + ordered stream compaction, zoom selection, divide_region and the
// 10-px-grid _sift_dup.  All of it is HBM/latency-bound integer and f64 work (no MFMA);
// built with -ffp-contract=off so every f64/f32 expression rounds once per operation,
// in the reference's operation order.
#include "az_geom_dev.h"

namespace {

constexpr int TB = 256;

// ----------------------------------------------------------------------------------------
__global__ void k_init_root(AzCounts *cnt, double *B0, int im_h, int im_w)
{
    // clears the counters of the previous search (one launch instead of a memset plus this)
    int *w = reinterpret_cast<int *>(cnt);
    for (int i = threadIdx.x; i < (int)(sizeof(AzCounts) / sizeof(int)); i += blockDim.x) w[i] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        // lib/detect/test.py:355: B = [[0, 0, W - 1.0, H - 1.0]]
        cnt->P[0] = 1;
        B0[0] = 0.0; B0[1] = 0.0; B0[2] = im_w - 1.0; B0[3] = im_h - 1.0;
    }
}

// lib/detect/test.py:61-97 (_get_rois_blob: f64 box * scale -> f32) and :212-214 (hash of
// np.round(rois * DEDUP_BOXES) . [1,1e3,1e6,1e9,1e12]; exact integers, so int64 here).
__global__ void k_rois_keys(const double *__restrict__ B, const int *Pptr, double scale, float dedup,
                            int batch, float *rois, long long *key, int *grp)
{
    const int P = *Pptr;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < P; r += gridDim.x * blockDim.x) {
        key[r] = roi_and_key(B + 4 * (size_t)r, scale, dedup, rois + 5 * (size_t)r, r);
        grp[r] = r / batch;                        // dedup is per BATCH_SIZE chunk (test.py:195-218)
    }
}

// first[i] = no j < i carries the same (grp, key): np.unique(return_index=True) keeps the
// first occurrence.  One wave per element, lanes stride over j (coalesced key reads served
// by L2), wave vote at the end: O(N^2 / 64) wave-steps, N is a few thousand at most.
__global__ void k_first(const long long *__restrict__ key, const int *__restrict__ grp, const int *Nptr,
                        unsigned char *first)
{
    const int N = *Nptr;
    const int lane = lane_id();
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < N; i += nwaves) {
        const long long ki = key[i];
        const int gi = grp ? grp[i] : 0;
        bool dup = false;
        for (int j0 = 0; j0 < i; j0 += 64) {
            const int j = j0 + lane;
            if (j < i) dup |= (key[j] == ki) & ((grp ? grp[j] : 0) == gi);
            if (__any(dup)) break;
        }
        const bool any_dup = __any(dup);            // vote with all lanes active
        if (lane == 0) first[i] = any_dup ? 0 : 1;
    }
}

// k_rois_keys and k_first in one launch for the roi dedup: a wave derives the keys it compares against
// from the regions themselves (a key is four multiplies and roundings), so there is no grid-wide
// dependency on a key array; it also stores its own element's roi / key / chunk id for k_dedup_rois.
__global__ void k_first_rois(const double *__restrict__ B, const int *Pptr, double scale, float dedup, int batch,
                             float *rois, long long *key, int *grp, unsigned char *first)
{
    const int P = *Pptr;
    const int lane = lane_id();
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < P; i += nwaves) {
        float roi5[5];
        const long long ki = roi_and_key(B + 4 * (size_t)i, scale, dedup, roi5, i);
        const int gi = i / batch;                      // dedup is per BATCH_SIZE chunk (test.py:195-218)
        if (lane == 0) { key[i] = ki; grp[i] = gi; }
        if (lane < 5) rois[5 * (size_t)i + lane] = roi5[lane];
        bool dup = false;
        for (int j0 = gi * batch; j0 < i; j0 += 64) {  // only the same chunk can hold a duplicate
            const int j = j0 + lane;
            if (j < i) {
                float r5[5];
                dup |= (roi_and_key(B + 4 * (size_t)j, scale, dedup, r5, j) == ki);
            }
            if (__any(dup)) break;
        }
        const bool any_dup = __any(dup);               // vote with all lanes active
        if (lane == 0) first[i] = any_dup ? 0 : 1;
    }
}

// slot[i] = number of distinct (grp, key) pairs ordered before i's pair = position of i's
// pair in np.unique's ascending output.  Same wave-per-element scheme, ballot + popcount.
__device__ __forceinline__ int dedup_slot(const long long *__restrict__ key, const int *__restrict__ grp,
                                          const unsigned char *__restrict__ first, int N, int i)
{
    const int lane = lane_id();
    const long long ki = key[i];
    const int gi = grp ? grp[i] : 0;
    int slot = 0;
    for (int j0 = 0; j0 < N; j0 += 64) {
        const int j = j0 + lane;
        bool c = false;
        if (j < N) {
            const int gj = grp ? grp[j] : 0;
            const long long kj = key[j];
            c = (first[j] != 0) & ((gj < gi) | ((gj == gi) & (kj < ki)));
        }
        slot += __popcll(__ballot(c));
    }
    return slot;
}

// Feature-space dedup outputs (lib/detect/test.py:215-218): index, inv_index and the
// gathered unique rois / anchor boxes (`boxes = boxes[index, :]`).
__global__ void k_dedup_rois(const long long *__restrict__ key, const int *__restrict__ grp, const int *Nptr,
                             const unsigned char *__restrict__ first, const float *__restrict__ rois,
                             const double *__restrict__ B, int *index, int *inv, float *urois, double *ubox,
                             int *Uptr)
{
    const int N = *Nptr;
    const int lane = lane_id();
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    int nfirst = 0;
    for (int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < N; i += nwaves) {
        const int slot = dedup_slot(key, grp, first, N, i);
        const bool f = first[i] != 0;
        if (lane == 0) inv[i] = slot;
        if (f) {
            if (lane == 0) { index[slot] = i; ++nfirst; }
            if (lane < 5) urois[5 * slot + lane] = rois[5 * i + lane];
            if (B && lane >= 8 && lane < 12) ubox[4 * slot + (lane - 8)] = B[4 * i + (lane - 8)];
        }
    }
    if (lane == 0 && nfirst) atomicAdd(Uptr, nfirst);
}

// _sift_dup output (lib/utils/div.pyx:85-89): regions[index] in ascending hash order.
__global__ void k_dedup_regions(const long long *__restrict__ key, const int *Nptr, int capOut,
                                const unsigned char *__restrict__ first, const double *__restrict__ child,
                                double *Bnext, int *Pnext, int *err, const int *__restrict__ csrc, int *srcnext)
{
    const int N = *Nptr;
    const int lane = lane_id();
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    int nfirst = 0;
    for (int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < N; i += nwaves) {
        if (!first[i]) continue;                     // wave-uniform
        const int slot = dedup_slot(key, nullptr, first, N, i);
        if (slot >= capOut) { if (lane == 0) atomicOr(err, 1); continue; }
        if (lane < 4) Bnext[4 * slot + lane] = child[4 * i + lane];
        if (lane == 0) { ++nfirst; if (csrc) srcnext[slot] = csrc[i]; }
    }
    if (lane == 0 && nfirst) atomicAdd(Pnext, nfirst);
}

// ----------------------------------------------------------------------------------------
// Candidate filter (test.py:171-187: keep min(w, h) + 1 >= MIN_SIDE, order r*11+s) and zoom
// selection (test.py:383-387: zoom[0] = 1 at level 1; indZ = where(zoom >= Tz), f32 zoom
// widened to double).  Pass 1: flags + per-block counts.
__global__ void k_flags(const AzCounts *cnt, int level, const int *__restrict__ inv,
                        const double *__restrict__ pred_u, const float *__restrict__ zoom_u, double Tz,
                        double min_side, int force_root, unsigned char *cflag, unsigned char *zflag,
                        int *bc_c, int *bc_z)
{
    const int P = cnt->P[level];
    const int NCAND = P * AZ_NSUB;
    const int nblk = (NCAND + TB - 1) / TB;
    for (int b = blockIdx.x; b < nblk; b += gridDim.x) {
        const int c = b * TB + threadIdx.x;
        int fl = 0;
        if (c < NCAND) {
            const int r = c / AZ_NSUB, s = c - r * AZ_NSUB;
            fl = cand_keep(pred_u + ((size_t)inv[r] * AZ_NSUB + s) * 4, min_side);
            cflag[c] = (unsigned char)fl;
        }
        int zf = 0;
        if (c < P) {
            float z = zoom_u[inv[c]];
            if (force_root && c == 0) z = 1.0f;
            zf = ((double)z >= Tz);
            zflag[c] = (unsigned char)zf;
        }
        const int n1 = __syncthreads_count(fl);
        const int n2 = __syncthreads_count(zf);
        if (threadIdx.x == 0) { bc_c[b] = n1; bc_z[b] = n2; }
    }
}

__device__ __forceinline__ int block_sum_upto(const int *v, int n, int *red)
{
    int s = 0;
    for (int t = threadIdx.x; t < n; t += blockDim.x) s += v[t];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
    __syncthreads();
    if (lane_id() == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    int tot = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += red[w];
    return tot;
}

// Pass 2: ordered compaction.  Candidates are appended to Y / aScores (test.py:380-381) and
// the zoom set Z = B[indZ] is gathered (test.py:387).
__global__ void k_compact(AzCounts *cnt, int level, int capCand, const double *__restrict__ B,
                          const int *__restrict__ inv, const double *__restrict__ pred_u,
                          const float *__restrict__ score_u, const unsigned char *__restrict__ cflag,
                          const unsigned char *__restrict__ zflag, const int *__restrict__ bc_c,
                          const int *__restrict__ bc_z, double *Yall, float *Sall, double *Z, int *zr)
{
    __shared__ int red[TB / 64];
    __shared__ int wsum[17];
    const int P = cnt->P[level];
    const int NCAND = P * AZ_NSUB;
    const int nblk = (NCAND + TB - 1) / TB;
    const int ybase = cnt->ytot[level];
    for (int b = blockIdx.x; b < nblk; b += gridDim.x) {
        const int base_c = block_sum_upto(bc_c, b, red);
        const int base_z = block_sum_upto(bc_z, b, red);
        const int c = b * TB + threadIdx.x;
        const int fl = (c < NCAND) ? cflag[c] : 0;
        int tot;
        const int off = block_excl_scan(fl, &tot, wsum);
        if (fl) {
            const int r = c / AZ_NSUB, s = c - r * AZ_NSUB;
            const size_t src = (size_t)inv[r] * AZ_NSUB + s;
            const int dst = ybase + base_c + off;
            if (dst < capCand) {
#pragma unroll
                for (int k = 0; k < 4; ++k) Yall[(size_t)dst * 4 + k] = pred_u[src * 4 + k];
                Sall[dst] = score_u[src];
            }
        }
        const int zf = (c < P) ? zflag[c] : 0;
        const int zoff = block_excl_scan(zf, &tot, wsum);
        if (zf) {
            const int dst = base_z + zoff;
#pragma unroll
            for (int k = 0; k < 4; ++k) Z[(size_t)dst * 4 + k] = B[(size_t)c * 4 + k];
            zr[dst] = c;
        }
    }
    if (blockIdx.x == 0) {
        const int tc = block_sum_upto(bc_c, nblk, red);
        const int tz = block_sum_upto(bc_z, nblk, red);
        if (threadIdx.x == 0) {
            int nc = tc;
            if (ybase + nc > capCand) { nc = capCand - ybase; atomicOr(&cnt->err, 2); }
            cnt->NC[level] = nc;
            cnt->ytot[level + 1] = ybase + nc;
            cnt->PZ[level] = tz;
        }
    }
}

// ----------------------------------------------------------------------------------------
// divide_region of every parent in ONE single-workgroup launch: children per parent + exclusive scan ->
// child offsets (total in *CHptr), then the children themselves (a few hundred parents at most on this
// path: a thread emits the <= 11 children of its parents).  With `src_off` the children also get a
// provenance id *src_base + src_add + src_off[zr ? zr[z] : z] + child index (speculative levels).
__global__ void __launch_bounds__(1024) k_divide(const int *PZptr, int *CHptr, int *err, int capCh,
                                                 const double *__restrict__ Z, int *choff, double min_side,
                                                 double *child, long long *ckey, const int *__restrict__ src_off,
                                                 const int *__restrict__ zr, const int *src_base, int src_add,
                                                 int *csrc)
{
    __shared__ int wsum[17];
    const int PZ = *PZptr;
    int running = 0;
    for (int base = 0; base < PZ; base += blockDim.x) {
        const int z = base + threadIdx.x;
        int n = 0;
        if (z < PZ) {
            const DivPlan p = div_plan(Z + 4 * (size_t)z);
            n = div_nchildren(p);
        }
        int tot;
        const int ex = block_excl_scan(n, &tot, wsum);
        if (z < PZ) choff[z] = running + ex;
        running += tot;
    }
    if (running > capCh) {
        if (threadIdx.x == 0) { atomicOr(err, 4); *CHptr = 0; }
        return;
    }
    if (threadIdx.x == 0) *CHptr = running;
    if (running == 0) return;
    __syncthreads();                                   // choff[] of other threads' parents
    for (int z = threadIdx.x; z < PZ; z += blockDim.x) {
        const double *r = Z + 4 * (size_t)z;
        const DivPlan p = div_plan(r);
        if (!p.num_long) continue;
        const int nb = div_nchildren(p);
        const size_t o = (size_t)choff[z];
        const int sbase = src_off ? (*src_base + src_add + src_off[zr ? zr[z] : z]) : 0;
        for (int bi = 0; bi < nb; ++bi) {
            double c[4];
            ckey[o + bi] = div_child(r, p, bi, min_side, c);
#pragma unroll
            for (int q = 0; q < 4; ++q) child[(o + bi) * 4 + q] = c[q];
            if (csrc) csrc[o + bi] = sbase + bi;
        }
    }
}

// Keys only (unit entry point az_sift_dup).
__global__ void k_region_keys(const double *__restrict__ regions, const int *Nptr, double min_side,
                              long long *ckey)
{
    const int N = *Nptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        long long h = 0, mult = 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            h += (long long)rint(regions[4 * (size_t)i + q] / min_side) * mult;
            mult *= 1000;
        }
        ckey[i] = h;
    }
}

}  // namespace

namespace {
__global__ void k_decode_unit(const double *__restrict__ anchors, const float *__restrict__ deltas,
                              const float *__restrict__ scores, int R, int im_h, int im_w, double eps,
                              double *pred_u, float *score_u)
{
    const int n = R * AZ_NSUB;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
        const int r = c / AZ_NSUB, s = c - r * AZ_NSUB;
        az_decode_box(anchors + 4 * (size_t)r, deltas + (size_t)r * 4 * AZ_NSUB + 4 * s, im_h, im_w, eps,
                      pred_u + 4 * (size_t)c);
        score_u[c] = scores[c];
    }
}
}  // namespace

// ----------------------------------------------------------------------------------------
static inline int grid_for(int cap, int per) { int g = (cap + per - 1) / per; return g < 1 ? 1 : (g > 2048 ? 2048 : g); }

void azk_init_root(hipStream_t s, AzCounts *cnt, double *B0, int im_h, int im_w)
{
    hipLaunchKernelGGL(k_init_root, dim3(1), dim3(64), 0, s, cnt, B0, im_h, im_w);
}

void azk_rois_keys(hipStream_t s, const double *B, const int *Pptr, int cap, double scale, float dedup,
                   int batch, float *rois, long long *key, int *grp)
{
    hipLaunchKernelGGL(k_rois_keys, dim3(grid_for(cap, TB)), dim3(TB), 0, s, B, Pptr, scale, dedup, batch,
                       rois, key, grp);
}

// roi projection + feature-space dedup of one level in two launches (keys + first occurrences, then slots)
void azk_rois_dedup(hipStream_t s, const double *B, const int *Pptr, int cap, double scale, float dedup, int batch,
                    float *rois, long long *key, int *grp, unsigned char *first, int *index, int *inv, float *urois,
                    double *ubox, int *Uptr)
{
    const int g = grid_for(cap, TB / 64);      // one wave per element
    hipLaunchKernelGGL(k_first_rois, dim3(g), dim3(TB), 0, s, B, Pptr, scale, dedup, batch, rois, key, grp, first);
    hipLaunchKernelGGL(k_dedup_rois, dim3(g), dim3(TB), 0, s, key, grp, Pptr, first, rois, B, index, inv,
                       urois, ubox, Uptr);
}

void azk_dedup_rois(hipStream_t s, const long long *key, const int *grp, const int *Nptr, int cap,
                    unsigned char *first, const float *rois, const double *B, int *index, int *inv,
                    float *urois, double *ubox, int *Uptr)
{
    const int g = grid_for(cap, TB / 64);      // one wave per element
    hipLaunchKernelGGL(k_first, dim3(g), dim3(TB), 0, s, key, grp, Nptr, first);
    hipLaunchKernelGGL(k_dedup_rois, dim3(g), dim3(TB), 0, s, key, grp, Nptr, first, rois, B, index, inv,
                       urois, ubox, Uptr);
}

void azk_flags_compact(hipStream_t s, AzCounts *cnt, int level, int capR, int capCand, const double *B,
                       const int *inv, const double *pred_u, const float *score_u, const float *zoom_u,
                       double Tz, double min_side, int force_root, unsigned char *cflag,
                       unsigned char *zflag, int *bc_c, int *bc_z, double *Yall, float *Sall, double *Z, int *zr)
{
    const int g = grid_for(capR * AZ_NSUB, TB);
    hipLaunchKernelGGL(k_flags, dim3(g), dim3(TB), 0, s, cnt, level, inv, pred_u, zoom_u, Tz, min_side,
                       force_root, cflag, zflag, bc_c, bc_z);
    hipLaunchKernelGGL(k_compact, dim3(g), dim3(TB), 0, s, cnt, level, capCand, B, inv, pred_u, score_u,
                       cflag, zflag, bc_c, bc_z, Yall, Sall, Z, zr);
}

void azk_divide(hipStream_t s, const int *PZptr, int *CHptr, int *err, int capR, int capCh, const double *Z,
                double min_side, int *choff, double *child, long long *ckey, const int *src_off, const int *zr,
                const int *src_base, int src_add, int *csrc)
{
    (void)capR;
    hipLaunchKernelGGL(k_divide, dim3(1), dim3(1024), 0, s, PZptr, CHptr, err, capCh, Z, choff, min_side, child, ckey,
                       src_off, zr, src_base, src_add, csrc);
}

void azk_dedup_regions(hipStream_t s, const long long *key, const int *Nptr, int cap, int capOut,
                       unsigned char *first, const double *child, double *Bnext, int *Pnext, int *err,
                       const int *csrc, int *srcnext)
{
    const int g = grid_for(cap, TB / 64);      // one wave per element
    hipLaunchKernelGGL(k_first, dim3(g), dim3(TB), 0, s, key, (const int *)nullptr, Nptr, first);
    hipLaunchKernelGGL(k_dedup_regions, dim3(g), dim3(TB), 0, s, key, Nptr, capOut, first, child, Bnext,
                       Pnext, err, csrc, srcnext);
}

void azk_region_keys(hipStream_t s, const double *regions, const int *Nptr, int cap, double min_side,
                     long long *ckey)
{
    hipLaunchKernelGGL(k_region_keys, dim3(grid_for(cap, TB)), dim3(TB), 0, s, regions, Nptr, min_side, ckey);
}

void azk_decode_unit(hipStream_t s, const double *anchors, const float *deltas, const float *scores, int R,
                     int im_h, int im_w, double eps, double *pred_u, float *score_u)
{
    hipLaunchKernelGGL(k_decode_unit, dim3(grid_for(R * AZ_NSUB, TB)), dim3(TB), 0, s, anchors, deltas,
                       scores, R, im_h, im_w, eps, pred_u, score_u);
}

namespace {
// Speculative rows S = [root ; B1 ; all children of all of B1] -> rois (no dedup): row 0 the
// root, rows 1..P1 = B1 in order, then the children in (parent, child) order.
__global__ void k_spec_rois(const double *__restrict__ root, const double *__restrict__ B1,
                            const double *__restrict__ C2, AzCounts *cnt, int capR, double scale, float *urois)
{
    const int P1 = cnt->specP1, CH = cnt->specCH;
    int total = 1 + P1 + CH;
    if (total > capR) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&cnt->err, 1); total = capR; }
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt->specU = total;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const double *b = (i == 0) ? root : (i <= P1 ? B1 + 4 * (size_t)(i - 1) : C2 + 4 * (size_t)(i - 1 - P1));
        urois[5 * (size_t)i] = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) urois[5 * (size_t)i + 1 + q] = (float)(b[q] * scale);
    }
}

// Head outputs of a speculative level: every unique roi's representative region r = index[u]
// has a row in the speculative pass (level 0: row 0; level 1: 1 + r; level 2: src2[r]); copy its
// scores and decode its deltas against the representative's own box.
__global__ void k_spec_lookup(int level, const int *Uptr, const int *__restrict__ index,
                              const int *__restrict__ src2, const double *__restrict__ ubox,
                              const float *__restrict__ zoom_s, const float *__restrict__ score_s,
                              const float *__restrict__ delta_s, int im_h, int im_w, double eps, float *zoom_u,
                              float *score_u, float *delta_u, double *pred_u)
{
    const int U = *Uptr;
    const int total = U * (AZ_NSUB + 1);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int u = idx / (AZ_NSUB + 1), t = idx - u * (AZ_NSUB + 1);
        const int r = index[u];
        const int srow = level == 0 ? 0 : (level == 1 ? 1 + r : src2[r]);
        if (t < AZ_NSUB) {
            score_u[(size_t)u * AZ_NSUB + t] = score_s[(size_t)srow * AZ_NSUB + t];
            float d4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                d4[q] = delta_s[(size_t)srow * 4 * AZ_NSUB + 4 * t + q];
                delta_u[(size_t)u * 4 * AZ_NSUB + 4 * t + q] = d4[q];
            }
            az_decode_box(ubox + 4 * (size_t)u, d4, im_h, im_w, eps, pred_u + ((size_t)u * AZ_NSUB + t) * 4);
        } else {
            zoom_u[u] = zoom_s[srow];
        }
    }
}
}  // namespace

void azk_spec_rois(hipStream_t s, const double *root, const double *B1, const double *C2, AzCounts *cnt, int capR,
                   double scale, float *urois)
{
    hipLaunchKernelGGL(k_spec_rois, dim3(8), dim3(TB), 0, s, root, B1, C2, cnt, capR, scale, urois);
}

void azk_spec_lookup(hipStream_t s, int level, const int *Uptr, const int *index, const int *src2,
                     const double *ubox, const float *zoom_s, const float *score_s, const float *delta_s, int im_h,
                     int im_w, double eps, float *zoom_u, float *score_u, float *delta_u, double *pred_u)
{
    hipLaunchKernelGGL(k_spec_lookup, dim3(64), dim3(TB), 0, s, level, Uptr, index, src2, ubox, zoom_s, score_s,
                       delta_s, im_h, im_w, eps, zoom_u, score_u, delta_u, pred_u);
}
