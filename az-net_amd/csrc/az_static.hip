// The search when the zoom test cannot fail (Tz <= 0).
//
// lib/detect/test.py:383-390 grows the tree with `indZ = np.where(zoom >= Tz)`; the zoom indicator is a Sigmoid
// output (test_fc.prototxt:221-232), so for Tz <= 0 -- the reference's TRAIN-phase setting, config.py:275 -- every
// region with a finite score zooms and B(level l+1) = divide_region(B(level l)) is a function of the image shape
// alone.  The regions of ALL levels, their rois, the 1/16 dedup maps and the anchors are then computed once per
// image shape (the same geometry kernels as the level loop, az_search.hip: ensure_static_plan) and every image runs
//   ONE head pass over the unique rois of all levels (RoIPool, int6, int7, tail), then
//   k_static_select (fixed proposal count) -- candidates of all levels appended in the reference's order, the
//   per-level counters and the final top-k in one launch -- or k_static_candidates + the selection kernels
// instead of one head pass + geometry per level: same rows through the same arithmetic (a roi's bits do not depend
// on which launch it sits in), no level-to-level dependency left.  The counters workgroup also verifies the premise
// (every zoom score of the tree >= Tz, i.e. no NaN): if it fails the host reruns the level loop.
// The level loop's LAST level uses the same construction (k_final_select at the end of this file).
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

// cand_src[c] = reg_u[c / 11] * 11 + c % 11: where candidate slot c (region-major x 11) finds its decoded box,
// score and selection key in the head pass's per-row outputs
__global__ void k_plan_cands(const int *__restrict__ reg_u, int Rtot, int *__restrict__ cand_src)
{
    const int N = Rtot * AZ_NSUB;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < N; c += gridDim.x * blockDim.x) {
        const int r = c / AZ_NSUB;
        cand_src[c] = reg_u[r] * AZ_NSUB + (c - r * AZ_NSUB);
    }
}

constexpr int SC_NT = 1024;
constexpr int SC_B = 8;                      // independent loads in flight per thread

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

// keys of slots tid + 1024*(m0 .. m0+7) (0 past the end): 8 + 8 independent loads, two memory round trips
__device__ __forceinline__ void load_keys8(const AzStaticArgs &a, int Nv, int m0, int tid, unsigned (&key)[SC_B])
{
    int src[SC_B];
#pragma unroll
    for (int j = 0; j < SC_B; ++j) src[j] = a.cand_src[min((m0 + j) * SC_NT + tid, Nv - 1)];
#pragma unroll
    for (int j = 0; j < SC_B; ++j) key[j] = a.key_u[src[j]];
#pragma unroll
    for (int j = 0; j < SC_B; ++j) key[j] = ((m0 + j) * SC_NT + tid < Nv) ? key[j] : 0u;
}

// Candidates of all levels, level-major, region-major, sub-region order (test.py:171-187, 380-381): workgroup b
// owns candidate slots [1024 b, 1024 b + 1024); its output offset is the number of kept candidates in all earlier
// slots, which it counts itself from the keys (a few thousand words, L2-resident) -- no inter-workgroup hand-off.
__device__ __forceinline__ void copy_role(const AzStaticArgs &a, int bid)
{
    __shared__ int red[16];
    __shared__ int wsum[17];
    const int tid = threadIdx.x;
    const int Nv = a.roff[a.nlev] * AZ_NSUB;
    int before = 0;
    for (int m0 = 0; m0 < bid; m0 += SC_B) {
        unsigned key[SC_B];
        load_keys8(a, Nv, m0, tid, key);
#pragma unroll
        for (int j = 0; j < SC_B; ++j) before += (m0 + j < bid && key[j] != 0u) ? 1 : 0;
    }
    const int base = block_sum(before, red);
    const int c = bid * SC_NT + tid;
    int fl = 0, src = 0;
    if (c < Nv) {
        src = a.cand_src[c];
        fl = a.key_u[src] != 0u;
    }
    int tot;
    const int off = block_excl_scan(fl, &tot, wsum);
    if (fl) {
        const int dst = base + off;
        if (dst < a.capCand) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a.Yall[(size_t)dst * 4 + k] = a.pred_u[(size_t)src * 4 + k];
            a.Sall[dst] = a.score_u[src];
        }
    }
}

// One workgroup writes every counter of the search (nothing else touches AzCounts in this launch) and checks the
// premise: indZ = where(zoom >= Tz) selects every region (the root is forced, test.py:383-384).
__device__ __forceinline__ void counters_role(const AzStaticArgs &a, int nsel_k)
{
    __shared__ int lev[16][AZ_MAX_LEVELS];
    __shared__ int sroff[AZ_MAX_LEVELS + 1];
    __shared__ int snc[AZ_MAX_LEVELS];
    __shared__ int sbad;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Rtot = a.roff[a.nlev];
    const int Nv = Rtot * AZ_NSUB;
    if (tid == 0) sbad = 0;
    if (tid <= AZ_MAX_LEVELS) sroff[tid] = tid <= a.nlev ? a.roff[tid] : 0x7fffffff;
    for (int i = tid; i < 16 * AZ_MAX_LEVELS; i += SC_NT) (&lev[0][0])[i] = 0;
    int bad = 0;
    for (int q = 1 + tid; q < Rtot; q += SC_NT) bad |= !((double)a.zoom_u[a.reg_u[q]] >= a.Tz);
    __syncthreads();
    if (bad) sbad = 1;
    // kept candidates per level: a wave's 64 slots are consecutive, so they span levels l(first) .. l(last)
    auto level_of = [&](int slot) {
        const int r = slot / AZ_NSUB;
        int l = 0;
#pragma unroll
        for (int q = 1; q < AZ_MAX_LEVELS; ++q) l += (r >= sroff[q]) ? 1 : 0;
        return min(l, a.nlev - 1);
    };
    for (int m0 = 0; m0 * SC_NT < Nv; m0 += SC_B) {
        unsigned key[SC_B];
        load_keys8(a, Nv, m0, tid, key);
#pragma unroll
        for (int j = 0; j < SC_B; ++j) {
            const int slot = (m0 + j) * SC_NT + tid;
            const int mine = level_of(min(slot, Nv - 1));
            const int lo = __builtin_amdgcn_readfirstlane(mine);
            const int hi = __builtin_amdgcn_readlane(mine, 63);
            for (int l = lo; l <= hi; ++l) {
                const int n = __popcll(__ballot(key[j] != 0u && mine == l));
                if (lane == 0) lev[wave][l] += n;
            }
        }
    }
    int *ci = reinterpret_cast<int *>(a.cnt);
    for (int i = tid; i < (int)(sizeof(AzCounts) / sizeof(int)); i += SC_NT) ci[i] = 0;
    __syncthreads();
    if (tid < a.nlev) {
        int nc = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) nc += lev[w][tid];
        snc[tid] = nc;
    }
    __syncthreads();
    if (tid == 0) {
        AzCounts *c = a.cnt;
        int err = sbad ? 32 : 0, y = 0;
        for (int l = 0; l < a.nlev; ++l) {
            int nc = snc[l];
            if (y + nc > a.capCand) { nc = a.capCand - y; err |= 2; }
            c->P[l] = sroff[l + 1] - sroff[l];
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
        if (nsel_k > 0) c->nsel = nsel_k < y ? nsel_k : y;
    }
}

// workgroups [0, nb): candidate writers; workgroup nb: the counters
__global__ void __launch_bounds__(SC_NT) k_static_candidates(AzStaticArgs a, int nb)
{
    if ((int)blockIdx.x < nb) copy_role(a, blockIdx.x);
    else counters_role(a, 0);
}

// The same launch also makes the final selection (test.py:396-400: argsort(-aScores)[:k]; ties: lower candidate
// index first, as az_select.hip).  Workgroups [0, nbC) are the candidate writers, workgroup nbC the counters,
// workgroup nbC + 1 + g ranks the 32 candidate slots [32g, 32g+32) of the UNCOMPACTED list (a dropped candidate has
// key 0 and never counts): rank(i) = #{j: key_j > key_i} + #{j < i: key_j == key_i} -- slot order is candidate
// order, so the ranks are those of az_select.hip's k_rank_count on the compacted list.
// Wave w of a ranking workgroup holds the w-th sixteenth of all keys in registers, one key per lane and register;
// for each own slot i the key k_i is a scalar and ONE v_cmp + s_bcnt1 counts 64 keys against it (a ballot).  Keys of
// 64-slot ranges before the own slots' range are compared as k_j + 1 > k_i (= k_j >= k_i: ties count), later ones
// as k_j > k_i; the one register that holds the own slots adds its ties among the lower lanes.
constexpr int SEL_I = 32;                    // own slots per ranking workgroup (255 workgroups for 8129 candidates)
constexpr int SEL_Q = 8;                     // keys per lane per pass (512 slots per wave per pass)

__global__ void __launch_bounds__(SC_NT) k_static_select(AzStaticArgs a, int nbC)
{
    if ((int)blockIdx.x < nbC) { copy_role(a, blockIdx.x); return; }
    if ((int)blockIdx.x == nbC) { counters_role(a, a.k); return; }
    __shared__ int part[16][SEL_I];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Nv = a.roff[a.nlev] * AZ_NSUB;
    const int Np = (Nv + SC_NT - 1) / SC_NT * SC_NT;      // slots, padded (padding keys are 0)
    const int jl = Np / 16;                               // slots per wave: a multiple of 64
    const int jb = wave * jl;
    const int i0 = ((int)blockIdx.x - nbC - 1) * SEL_I;
    const int ibase = i0 & ~63, ioff = i0 - ibase;        // the 64-slot register range the own slots sit in
    const int i = i0 + lane;
    const int isrc = a.cand_src[min(i, Nv - 1)];
    const unsigned kiv = (lane < SEL_I && i < Nv) ? a.key_u[isrc] : 0u;     // lane l < 32: the key of own slot i0 + l
    int cntv = 0;                                         // lane l: earlier-sorting candidates found by this wave
    for (int p0 = 0; p0 < jl; p0 += SEL_Q * 64) {
        int src[SEL_Q];
        unsigned key[SEL_Q];
#pragma unroll
        for (int q = 0; q < SEL_Q; ++q) src[q] = a.cand_src[min(jb + p0 + q * 64 + lane, Nv - 1)];
#pragma unroll
        for (int q = 0; q < SEL_Q; ++q) key[q] = a.key_u[src[q]];
        unsigned long long own = 0;                       // (wave-uniform) ballot source of the register that is the own slots
        unsigned kown = 0;
#pragma unroll
        for (int q = 0; q < SEL_Q; ++q) {
            const int jq = jb + p0 + q * 64;              // wave-uniform
            const bool live = jq < jl + jb && jq + lane < Nv;
            key[q] = live ? key[q] : 0u;
            if (jq == ibase) { kown = key[q]; own = 1; }
            else if (jq < ibase) key[q] = key[q] == 0xFFFFFFFFu ? key[q] : key[q] + (key[q] ? 1u : 0u);
        }
        for (int ii = 0; ii < SEL_I; ++ii) {
            const unsigned ki = __builtin_amdgcn_readlane(kiv, ii);
            int n = 0;
#pragma unroll
            for (int q = 0; q < SEL_Q; ++q) n += __popcll(__ballot(key[q] > ki));
            if (own) n += __popcll(__ballot(kown == ki) & ((1ull << (ii + ioff)) - 1ull));
            cntv += (lane == ii) ? n : 0;
        }
    }
    if (lane < SEL_I) part[wave][lane] = cntv;
    __syncthreads();
    if (wave == 0 && lane < SEL_I && kiv != 0u) {
        int rank = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) rank += part[w][lane];
        if (rank < a.k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) a.Yout[(size_t)rank * 4 + q] = a.pred_u[(size_t)isrc * 4 + q];
            a.Sout[rank] = a.score_u[isrc];
        }
    }
}

// ---- the level loop's last level: candidates + counters + top-k in one launch -------------------------------------
// The list to rank is [candidates of the earlier levels, already in Yall / Sall] ++ [the last level's 11 P slots,
// found through inv]; list order is candidate order.  Roles as in k_static_select; all sizes are read on the device: a
// fixed grid whose workgroups take turns (48 writers, 512 rankers: one turn each up to 16 384 candidates).
__device__ __forceinline__ unsigned final_key(const AzFinalArgs &a, int prev, int Ns, int j)
{
    if (j < prev) { const unsigned k = score_key(a.Sall[j]); return k ? k : 1u; }
    const int c = j - prev;
    if (c >= Ns) return 0u;
    const int r = c / AZ_NSUB;
    return a.key_u[(size_t)a.inv[r] * AZ_NSUB + (c - r * AZ_NSUB)];
}

constexpr int FIN_NBC = 48;        // writer workgroups (1024 slots each per turn)
constexpr int FIN_NBR = 512;       // ranking workgroups (32 slots each per turn)

static __device__ __forceinline__ void final_select_body(const AzFinalArgs &a)
{
    constexpr int nbC = FIN_NBC;
    __shared__ int red[16];
    __shared__ int wsum[17];
    __shared__ int part[16][SEL_I];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.cnt->err & (8 | 2048)) return;   // a fused level (2048: a batch's pass) overflowed: its outputs are not there, the host reruns the search
    const int P = a.cnt->P[a.level], prev = a.cnt->ytot[a.level];
    const int Ns = P * AZ_NSUB, Nv = prev + Ns;
    if ((int)blockIdx.x < nbC) {
        // ---- writers: the last level's kept candidates go behind the earlier levels' (test.py:380-381)
        for (int bid = blockIdx.x; bid * SC_NT < Ns; bid += nbC) {
            // kept slots before this workgroup's range: the keys of 8 x 1024 slots per turn, their loads (inv, then the
            // key: two dependent round trips) issued together -- one slot range per turn cost two round trips EACH, and
            // the writer of the last range set the kernel's time
            int before = 0;
            for (int m0 = 0; m0 < bid; m0 += SC_B) {
                unsigned k8[SC_B];
#pragma unroll
                for (int j = 0; j < SC_B; ++j) k8[j] = final_key(a, 0, Ns, min((m0 + j) * SC_NT + tid, Ns - 1));
#pragma unroll
                for (int j = 0; j < SC_B; ++j) before += (m0 + j < bid && k8[j] != 0u) ? 1 : 0;
            }
            const int base = block_sum(before, red);
            const int c = bid * SC_NT + tid;
            int fl = 0, src = 0;
            if (c < Ns) {
                const int r = c / AZ_NSUB;
                src = a.inv[r] * AZ_NSUB + (c - r * AZ_NSUB);
                fl = a.key_u[src] != 0u;
            }
            int tot;
            const int off = block_excl_scan(fl, &tot, wsum);
            if (fl) {
                const int dst = prev + base + off;
                if (dst < a.capCand) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) a.Yall[(size_t)dst * 4 + q] = a.pred_u[(size_t)src * 4 + q];
                    a.Sall[dst] = a.score_u[src];
                }
            }
        }
        return;
    }
    if ((int)blockIdx.x == nbC) {
        // ---- the level's counters (test.py:383-387: zoom[0] = 1 at level 1; indZ = where(zoom >= Tz))
        int kept = 0, zoomed = 0;
        for (int c = tid; c < Ns; c += SC_NT) kept += final_key(a, 0, Ns, c) != 0u;
        for (int r = tid; r < P; r += SC_NT) {
            float z = a.zoom_u[a.inv[r]];
            if (a.force_root && r == 0) z = 1.0f;
            zoomed += ((double)z >= a.Tz) ? 1 : 0;
        }
        const int nc_all = block_sum(kept, red);
        const int nz = block_sum(zoomed, red);
        if (tid == 0) {
            int nc = nc_all;
            if (prev + nc > a.capCand) { nc = a.capCand - prev; atomicOr(&a.cnt->err, 2); }
            a.cnt->NC[a.level] = nc;
            a.cnt->ytot[a.level + 1] = prev + nc;
            a.cnt->PZ[a.level] = nz;
            a.cnt->nsel = a.k < prev + nc ? a.k : prev + nc;
        }
        return;
    }
    // ---- rankers (see k_static_select)
    const int Np = (Nv + SC_NT - 1) / SC_NT * SC_NT;
    const int jl = Np / 16, jb = wave * jl;
    for (int i0 = ((int)blockIdx.x - nbC - 1) * SEL_I; i0 < Nv; i0 += FIN_NBR * SEL_I) {
        const int ibase = i0 & ~63, ioff = i0 - ibase;
        const int i = i0 + lane;
        const unsigned kiv = (lane < SEL_I && i < Nv) ? final_key(a, prev, Ns, i) : 0u;
        int cntv = 0;
        for (int p0 = 0; p0 < jl; p0 += SEL_Q * 64) {
            unsigned key[SEL_Q];
#pragma unroll
            for (int q = 0; q < SEL_Q; ++q) {
                const int jq = jb + p0 + q * 64;
                key[q] = (jq < jb + jl && jq + lane < Nv) ? final_key(a, prev, Ns, jq + lane) : 0u;
            }
            unsigned long long own = 0;
            unsigned kown = 0;
#pragma unroll
            for (int q = 0; q < SEL_Q; ++q) {
                const int jq = jb + p0 + q * 64;
                if (jq == ibase) { kown = key[q]; own = 1; }
                else if (jq < ibase) key[q] = key[q] == 0xFFFFFFFFu ? key[q] : key[q] + (key[q] ? 1u : 0u);
            }
            for (int ii = 0; ii < SEL_I; ++ii) {
                const unsigned ki = __builtin_amdgcn_readlane(kiv, ii);
                int n = 0;
#pragma unroll
                for (int q = 0; q < SEL_Q; ++q) n += __popcll(__ballot(key[q] > ki));
                if (own) n += __popcll(__ballot(kown == ki) & ((1ull << (ii + ioff)) - 1ull));
                cntv += (lane == ii) ? n : 0;
            }
        }
        __syncthreads();             // (the previous turn's readers of `part` are done)
        if (lane < SEL_I) part[wave][lane] = cntv;
        __syncthreads();
        if (wave == 0 && lane < SEL_I && kiv != 0u) {
            int rank = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) rank += part[w][lane];
            if (rank < a.k) {
                if (i < prev) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) a.Yout[(size_t)rank * 4 + q] = a.Yall[(size_t)i * 4 + q];
                    a.Sout[rank] = a.Sall[i];
                } else {
                    const int c = i - prev, r = c / AZ_NSUB;
                    const size_t src = (size_t)a.inv[r] * AZ_NSUB + (c - r * AZ_NSUB);
#pragma unroll
                    for (int q = 0; q < 4; ++q) a.Yout[(size_t)rank * 4 + q] = a.pred_u[src * 4 + q];
                    a.Sout[rank] = a.score_u[src];
                }
            }
        }
    }
}

__global__ void __launch_bounds__(SC_NT) k_final_select(AzFinalArgs a) { final_select_body(a); }
// a batch of images searched in lockstep (az_batch.hip): workgroups (*, b) are image b's, its arguments in device memory
__global__ void __launch_bounds__(SC_NT) k_final_select_b(const AzFinalArgs *args) { AZ_UNIFORM_ARGS(AzFinalArgs, a, args + blockIdx.y); final_select_body(a); }

}  // namespace

void azk_plan_rows(hipStream_t s, const int *inv, const int *Pptr, int capR, int roff, int uoff, int *reg_u)
{
    k_plan_rows<<<dim3((capR + 255) / 256 > 64 ? 64 : (capR + 255) / 256), dim3(256), 0, s>>>(inv, Pptr, roff, uoff, reg_u);
}

bool azk_static_select(hipStream_t s, const AzStaticArgs &a)
{
    const int Nv = a.roff[a.nlev] * AZ_NSUB;
    if (a.k <= 0 || Nv <= 0) return false;
    const int nbC = (Nv + SC_NT - 1) / SC_NT;
    const int nbR = (Nv + SEL_I - 1) / SEL_I;
    k_static_select<<<dim3(nbC + 1 + nbR), dim3(SC_NT), 0, s>>>(a, nbC);
    return true;
}

void azk_final_select(hipStream_t s, const AzFinalArgs &a)
{
    k_final_select<<<dim3(FIN_NBC + 1 + FIN_NBR), dim3(SC_NT), 0, s>>>(a);
}

void azk_final_select_batch(hipStream_t s, const AzFinalArgs *args_dev, int n)
{
    k_final_select_b<<<dim3(FIN_NBC + 1 + FIN_NBR, n), dim3(SC_NT), 0, s>>>(args_dev);
}

void azk_plan_cands(hipStream_t s, const int *reg_u, int Rtot, int *cand_src)
{
    const int nb = (Rtot * AZ_NSUB + 255) / 256;
    k_plan_cands<<<dim3(nb > 256 ? 256 : (nb > 0 ? nb : 1)), dim3(256), 0, s>>>(reg_u, Rtot, cand_src);
}

void azk_static_candidates(hipStream_t s, const AzStaticArgs &a)
{
    const int Nv = a.roff[a.nlev] * AZ_NSUB;
    const int nb = (Nv + SC_NT - 1) / SC_NT;
    k_static_candidates<<<dim3(nb + 1), dim3(SC_NT), 0, s>>>(a, nb);
}

// ---- whole-tree speculation: the window table of an image shape's plan rows, and the speculative rows' map ----------
namespace {
__global__ void k_full_tab_build(const float *__restrict__ urois, int n_rows, int root_row, float ss,
                                 unsigned long long *tab, unsigned T, int *err)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    unsigned long long key;
    if (!pool_key(urois + 5 * (size_t)r, ss, &key)) { atomicOr(err, 1); return; }
    const unsigned long long word = (key << 13) | (unsigned)(r == root_row ? AZ_TAB_ROOT : (unsigned)r);
    unsigned slot = az_tab_hash(key, T);
    for (;;) {
        unsigned long long old = tab[slot];
        if (old == ~0ull) {
            old = atomicCAS(&tab[slot], ~0ull, word);
            if (old == ~0ull) return;
        }
        if ((old >> 13) == key) { atomicMin(&tab[slot], word); return; }       // (same window twice: either row will do)
        slot = slot + 1 == T ? 0u : slot + 1;
    }
}

// every row of the speculative pass (levels 1-3) -> its row in the whole-tree pass; a window the plan does not hold (a
// _sift_dup duplicate with other coordinates than the survivor) becomes an extra row behind the plan's non-root rows
__global__ void k_full_map(const float *__restrict__ spec_urois, int n_spec, float ss, const unsigned long long *tab,
                           unsigned T, int base_extra, int cap_rows, float *urois_full, double *ubox_full, int *map,
                           int *n_extra, int *err)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_spec) return;
    const float *roi = spec_urois + 5 * (size_t)i;
    int row = az_tab_lookup(tab, T, roi, ss, (int)AZ_TAB_ROOT);
    if (row < 0) {
        const int e = atomicAdd(n_extra, 1);
        row = base_extra + e;
        if (row + 1 >= cap_rows) { atomicOr(err, 1); map[i] = 0; return; }
        for (int q = 0; q < 5; ++q) urois_full[5 * (size_t)row + q] = roi[q];
        for (int q = 0; q < 4; ++q) ubox_full[4 * (size_t)row + q] = 0.0;          // (decoded boxes of these rows are never read)
    }
    map[i] = row;
}
}  // namespace

// ---- the closure rows: every region ANY pruning of the tree can produce ---------------------------------------------------
// B(l+1) = _sift_dup(divide_region(B(l)[zoom >= Tz])) (test.py:386-390, div.pyx:15-89): whatever the zoom scores, a region
// of level l+1 is a child of a region of level l, and _sift_dup only ever DROPS children (which of several same-hash
// children survives depends on which parents zoomed).  So C(0) = {root}, C(l+1) = all children of all of C(l) holds every
// region a search of this image shape can meet; one row per distinct RoIPool window among them serves every Tz.
namespace {
__global__ void k_closure_rois(const double *__restrict__ regs, int n, double scale, float *__restrict__ out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        out[5 * (size_t)i] = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) out[5 * (size_t)i + 1 + q] = (float)(regs[4 * (size_t)i + q] * scale);   // test.py:61-97
    }
}

// The table was built over ALL closure rois with word = (window << 13) | smallest index holding that window (index 0 = the
// root = AZ_TAB_ROOT).  Rows of the pass: the window owners other than the root, in index order.  One workgroup:
// newrow[i] = row of owner i (-1: not an owner), the owners' rois gathered; *n_rows = number of rows.
__global__ void __launch_bounds__(1024) k_closure_compact(const float *__restrict__ all, int N, float ss,
                                                           const unsigned long long *__restrict__ tab, unsigned T,
                                                           int *newrow, float *urois_full, double *ubox_full, int *n_rows)
{
    __shared__ int wsum[17];
    int run = 0;
    for (int base = 0; base < N; base += 1024) {
        const int i = base + (int)threadIdx.x;
        int head = 0;
        if (i > 0 && i < N) head = az_tab_lookup(tab, T, all + 5 * (size_t)i, ss, -2) == i;
        int tot;
        const int ex = block_excl_scan(head, &tot, wsum);
        if (i < N) newrow[i] = head ? run + ex : -1;
        if (head) {
            const size_t row = (size_t)(run + ex);
#pragma unroll
            for (int q = 0; q < 5; ++q) urois_full[5 * row + q] = all[5 * (size_t)i + q];
#pragma unroll
            for (int q = 0; q < 4; ++q) ubox_full[4 * row + q] = 0.0;      // (every level decodes against its own boxes)
        }
        run += tot;
    }
    if (threadIdx.x == 0) *n_rows = run;
}

__global__ void k_closure_relabel(unsigned long long *tab, unsigned T, const int *__restrict__ newrow, int *err)
{
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
        const unsigned long long w = tab[t];
        if (w == ~0ull) continue;
        const unsigned r = (unsigned)(w & 0x1FFFu);
        if (r == AZ_TAB_ROOT) continue;
        const int nr = newrow[r];
        if (nr < 0 || nr >= (int)AZ_TAB_ROOT) { atomicOr(err, 1); continue; }
        tab[t] = (w & ~0x1FFFull) | (unsigned)nr;
    }
}
}  // namespace

void azk_closure_rois(hipStream_t s, const double *regs, int n, double scale, float *out)
{
    if (n > 0) hipLaunchKernelGGL(k_closure_rois, dim3((n + 255) / 256), dim3(256), 0, s, regs, n, scale, out);
}

void azk_closure_compact(hipStream_t s, const float *all, int N, float ss, unsigned long long *tab, unsigned T, int *newrow,
                         float *urois_full, double *ubox_full, int *n_rows, int *err)
{
    hipLaunchKernelGGL(k_closure_compact, dim3(1), dim3(1024), 0, s, all, N, ss, tab, T, newrow, urois_full, ubox_full, n_rows);
    hipLaunchKernelGGL(k_closure_relabel, dim3((T + 255) / 256), dim3(256), 0, s, tab, T, newrow, err);
}

namespace {
// Whole-tree speculation, a level that runs on the multi-launch geometry kernels (more regions than the fused level kernel
// holds): the head outputs of its unique rois by RoIPool window among the rows of the search's one pass -- the lookup
// stage of k_level_geom, chip-wide.  ubox = the representative's own box of every unique roi (k_dedup_rois).
__global__ void __launch_bounds__(256) k_full_lookup(const int *Uptr, const float *__restrict__ urois,
                                                      const double *__restrict__ ubox, const unsigned long long *tab,
                                                      unsigned T, int root_row, float ss,
                                                      const float *__restrict__ delta_all, const float *__restrict__ score_all,
                                                      const float *__restrict__ zoom_all, int im_h, int im_w, double eps,
                                                      double min_side, double *pred_v, float *score_v, float *zoom_v,
                                                      unsigned char *keep_v, unsigned *key_v, int *err)
{
    const int U = *Uptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < U * AZ_NSUB; i += gridDim.x * blockDim.x) {
        const int slot = i / AZ_NSUB, sub = i - slot * AZ_NSUB;
        const int row = az_tab_lookup(tab, T, urois + 5 * (size_t)slot, ss, root_row);
        if (row < 0) { atomicOr(err, 8 | 256); continue; }
        float d4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) d4[q] = delta_all[(size_t)row * 4 * AZ_NSUB + 4 * sub + q];
        double bx[4];
        az_decode_box(ubox + 4 * (size_t)slot, d4, im_h, im_w, eps, bx);
#pragma unroll
        for (int q = 0; q < 4; ++q) pred_v[(size_t)i * 4 + q] = bx[q];
        const float sc = score_all[(size_t)row * AZ_NSUB + sub];
        score_v[i] = sc;
        const bool kp = cand_keep(bx, min_side);
        keep_v[i] = kp ? 1 : 0;
        const unsigned kk = score_key(sc);
        key_v[i] = kp ? (kk ? kk : 1u) : 0u;
        if (sub == 0) zoom_v[slot] = zoom_all[row];
    }
}
}  // namespace

void azk_full_lookup(hipStream_t s, const int *Uptr, const float *urois, const double *ubox, const unsigned long long *tab,
                     unsigned T, int root_row, float ss, const float *delta_all, const float *score_all, const float *zoom_all,
                     int im_h, int im_w, double eps, double min_side, double *pred_v, float *score_v, float *zoom_v,
                     unsigned char *keep_v, unsigned *key_v, int *err)
{
    hipLaunchKernelGGL(k_full_lookup, dim3(256), dim3(256), 0, s, Uptr, urois, ubox, tab, T, root_row, ss, delta_all, score_all,
                       zoom_all, im_h, im_w, eps, min_side, pred_v, score_v, zoom_v, keep_v, key_v, err);
}

void azk_full_tab_build(hipStream_t s, const float *urois, int n_rows, int root_row, float ss, unsigned long long *tab,
                        unsigned T, int *err)
{
    hipMemsetAsync(tab, 0xFF, (size_t)T * sizeof(unsigned long long), s);
    hipLaunchKernelGGL(k_full_tab_build, dim3((n_rows + 255) / 256), dim3(256), 0, s, urois, n_rows, root_row, ss, tab, T, err);
}

void azk_full_map(hipStream_t s, const float *spec_urois, int n_spec, float ss, const unsigned long long *tab, unsigned T,
                  int base_extra, int cap_rows, float *urois_full, double *ubox_full, int *map, int *n_extra, int *err)
{
    hipLaunchKernelGGL(k_full_map, dim3((n_spec + 255) / 256), dim3(256), 0, s, spec_urois, n_spec, ss, tab, T, base_extra,
                       cap_rows, urois_full, ubox_full, map, n_extra, err);
}
