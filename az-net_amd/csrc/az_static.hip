// The search when the zoom test cannot fail (Tz <= 0).
//
// lib/detect/test.py:383-390 grows the tree with `indZ = np.where(zoom >= Tz)`; the zoom indicator is a Sigmoid
// output (test_fc.prototxt:221-232), so for Tz <= 0 -- the reference's TRAIN-phase setting, config.py:275 -- every
// region with a finite score zooms and B(level l+1) = divide_region(B(level l)) is a function of the image shape
// alone.  The regions of ALL levels, their rois, the 1/16 dedup maps and the anchors are then computed once per
// image shape (the same geometry kernels as the level loop, az_capi.hip: ensure_static_plan) and every image runs
//   ONE head pass over the unique rois of all levels (RoIPool, int6, int7, tail), then
//   k_static_candidates: candidates of all levels appended in the reference's order + the per-level counters,
//   then the final selection
// instead of one head pass + geometry per level: same rows through the same arithmetic (a roi's bits do not depend
// on which launch it sits in), no level-to-level dependency left.  k_static_candidates also verifies the premise
// (every zoom score of the tree >= Tz, i.e. no NaN): if it fails the host reruns the level loop.
#include <hip/hip_runtime.h>
#include "az_dev.h"
#include "az_geom_dev.h"

namespace {

// reg_u[roff + r] = uoff + inv[r]: the row of the one head pass that serves region r of this level
__global__ void k_plan_rows(const int *__restrict__ inv, const int *Pptr, int roff, int uoff, int *__restrict__ reg_u)
{
    const int P = *Pptr;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < P; r += gridDim.x * blockDim.x) reg_u[roff + r] = uoff + inv[r];
}

constexpr int SC_NT = 1024;
constexpr int SC_RB = SC_NT / AZ_NSUB;       // regions per workgroup (93 -> 1023 candidates)

__device__ __forceinline__ int keep_count(const unsigned char *__restrict__ keep_u, int u)
{
    const unsigned char *k = keep_u + (size_t)u * AZ_NSUB;
    int n = 0;
#pragma unroll
    for (int t = 0; t < AZ_NSUB; ++t) n += k[t];
    return n;
}

__device__ __forceinline__ int block_sum(int v, int *red /* >= 16 ints */)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if (lane_id() == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    int tot = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += red[w];
    return tot;
}

// Candidates of all levels, level-major, region-major, sub-region order (test.py:171-187, 380-381): workgroup b
// owns regions [b*93, b*93+93); its output offset is the number of kept candidates of all earlier regions, which it
// counts itself from the keep flags (<= a few thousand bytes, L2-resident) -- no inter-workgroup hand-off.
// Workgroup 0 also writes every counter of the search and checks the premise.
__global__ void __launch_bounds__(SC_NT) k_static_candidates(AzStaticArgs a)
{
    __shared__ int red[16];
    __shared__ int wsum[17];
    __shared__ int lev[16][AZ_MAX_LEVELS];
    __shared__ int sbad;
    const int tid = threadIdx.x;
    const int Rtot = a.roff[a.nlev];
    const int r0 = blockIdx.x * SC_RB;

    int before = 0;
    for (int r = tid; r < r0; r += SC_NT) before += keep_count(a.keep_u, a.reg_u[r]);
    const int base = block_sum(before, red);

    const int rl = tid / AZ_NSUB, s = tid - rl * AZ_NSUB;
    const int r = r0 + rl;
    int fl = 0;
    size_t src = 0;
    if (rl < SC_RB && r < Rtot) {
        src = (size_t)a.reg_u[r] * AZ_NSUB + s;
        fl = a.keep_u[src];
    }
    int tot;
    const int off = block_excl_scan(fl, &tot, wsum);
    if (fl) {
        const int dst = base + off;
        if (dst < a.capCand) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a.Yall[(size_t)dst * 4 + k] = a.pred_u[src * 4 + k];
            a.Sall[dst] = a.score_u[src];
        }
    }
    if (blockIdx.x != 0) return;

    // ---- workgroup 0: the counters of the whole search --------------------------------------------------------
    if (tid == 0) sbad = 0;
    const int wave = tid >> 6;
    for (int l = 0; l < a.nlev; ++l) {
        int n = 0;
        for (int q = a.roff[l] + tid; q < a.roff[l + 1]; q += SC_NT) n += keep_count(a.keep_u, a.reg_u[q]);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) n += __shfl_xor(n, d, 64);
        if (lane_id() == 0) lev[wave][l] = n;
    }
    __syncthreads();
    // the premise: indZ = where(zoom >= Tz) selects every region (the root is forced, test.py:383-384)
    int bad = 0;
    for (int q = 1 + tid; q < Rtot; q += SC_NT) bad |= !((double)a.zoom_u[a.reg_u[q]] >= a.Tz);
    if (bad) sbad = 1;
    int *ci = reinterpret_cast<int *>(a.cnt);
    for (int i = tid; i < (int)(sizeof(AzCounts) / sizeof(int)); i += SC_NT) ci[i] = 0;
    __syncthreads();
    if (tid == 0) {
        AzCounts *c = a.cnt;
        int err = sbad ? 32 : 0, y = 0;
        for (int l = 0; l < a.nlev; ++l) {
            int nc = 0;
            for (int w = 0; w < 16; ++w) nc += lev[w][l];
            if (y + nc > a.capCand) { nc = a.capCand - y; err |= 2; }
            c->P[l] = a.roff[l + 1] - a.roff[l];
            c->U[l] = a.U[l];
            c->PZ[l] = c->P[l];
            c->CH[l] = a.CH[l];
            c->NC[l] = nc;
            c->ytot[l] = y;
            y += nc;
        }
        c->ytot[a.nlev] = y;
        c->specU = a.Utot;
        c->err = err;
    }
}

}  // namespace

void azk_plan_rows(hipStream_t s, const int *inv, const int *Pptr, int capR, int roff, int uoff, int *reg_u)
{
    k_plan_rows<<<dim3((capR + 255) / 256 > 64 ? 64 : (capR + 255) / 256), dim3(256), 0, s>>>(inv, Pptr, roff, uoff, reg_u);
}

void azk_static_candidates(hipStream_t s, const AzStaticArgs &a)
{
    const int Rtot = a.roff[a.nlev];
    const int nb = (Rtot + SC_RB - 1) / SC_RB;
    k_static_candidates<<<dim3(nb > 0 ? nb : 1), dim3(SC_NT), 0, s>>>(a);
}
