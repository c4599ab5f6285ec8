// az_batch.hip -- the two small kernels that let SEVERAL images walk their zoom trees in lockstep (az_search.hip:
// batch_launch_impl): the rois every image forwards at a level go through the head in ONE pass.
//
// The reference forwards one image at a time (lib/detect/test.py:508-513, one `_az_forward` per level of one image,
// test.py:373-391); its roi blob nevertheless carries Caffe's batch index in column 0 (test.py:93-97, always 0 there).
// At a tuned threshold a level of one image is a few dozen rois -- far too few for a pass over the 411 MB of int6 weights
// to be anything but a weight stream -- and what one image does at a level does not depend on any other image.  So B
// images (of one shape or of several, as long as their searches have the same number of levels) are searched together: every image keeps its own tree (an az_ctx of its own: regions, counters,
// candidates, geometry kernels as workgroups (., b) of one launch, az_fused.hip / az_level.hip / az_static.hip), and per
// level
//   k_batch_gather    concatenates the images' unique rois -- column 0 = the image's index in the batch, which RoIPool
//                     reads as Caffe's roi_batch_ind (az_head.hip) -- and their anchor boxes, and leaves the row offsets
//                     and the pass's row count on the device (no host synchronisation anywhere); also every image's
//                     map size (RoIPool clamps a window to ITS map) and every row's image size (the heads clip a decoded
//                     box to ITS image): the images of a batch may differ in shape;
//   the head          RoIPool, int6, slab sum, int7, heads: the unchanged kernels on the concatenated rows (a row's bits do
//                     not depend on which rows share its launch: tests/test_gpu_parity.py);
//   k_batch_scatter   hands every image its rows of the head's outputs, where its geometry kernel expects them.
// Same results as the level loop on each image alone, bit for bit (tests/test_gpu_batch.py).
#include <hip/hip_runtime.h>
#include "az_dev.h"

namespace {

// offsets of the images' rows in the pass; an image whose search has failed (error word set) forwards nothing more
__device__ __forceinline__ int batch_offsets(const AzGatherArgs &a, int *off /* [n + 1], LDS or registers */)
{
    int run = 0;
    for (int b = 0; b < a.n; ++b) {
        off[b] = run;
        int r = *a.rows[b];
        if ((a.err[b] && *a.err[b] != 0) || r < 0) r = 0;
        run += r;
    }
    off[a.n] = run;
    return run;
}

__global__ void __launch_bounds__(256) k_batch_gather(AzGatherArgs a)
{
    __shared__ int off[AZ_BATCH_MAX + 1];
    if (threadIdx.x == 0) batch_offsets(a, off);
    __syncthreads();
    const int total = off[a.n];
    if (total > a.capR) {
        // the pass does not fit the head's buffers: every image's search is marked and run again on its own (host)
        if (blockIdx.x == 0 && threadIdx.x < a.n && a.err[threadIdx.x]) atomicOr(a.err[threadIdx.x], 2048);
        if (blockIdx.x == 0 && threadIdx.x == 0) { for (int b = 0; b <= a.n; ++b) a.off_out[b] = 0; a.off_out[AZ_BATCH_MAX + 1] = 0; }
        return;
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x <= a.n) a.off_out[threadIdx.x] = off[threadIdx.x];
        if (threadIdx.x == 0) a.off_out[AZ_BATCH_MAX + 1] = total;          // the pass's row count (the head kernels' Mptr)
        if (threadIdx.x < a.n) {
            a.feats_out[threadIdx.x] = a.feat[threadIdx.x];
            a.feat_hw_out[2 * threadIdx.x] = a.fh[threadIdx.x];
            a.feat_hw_out[2 * threadIdx.x + 1] = a.fw[threadIdx.x];
        }
    }
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < total; r += gridDim.x * blockDim.x) {
        int b = 0;
        while (b + 1 < a.n && r >= off[b + 1]) ++b;
        const int i = r - off[b];
        const float *src = a.rois[b] + 5 * (size_t)i;
        float *dst = a.rois_cat + 5 * (size_t)r;
        dst[0] = (float)b;                                                   // Caffe's roi_batch_ind
        dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3]; dst[4] = src[4];
        a.row_hw_out[2 * (size_t)r] = a.im_h[b];
        a.row_hw_out[2 * (size_t)r + 1] = a.im_w[b];
        if (a.ubox[b]) {
            const double *ub = a.ubox[b] + 4 * (size_t)i;
            double *ud = a.ubox_cat + 4 * (size_t)r;
            ud[0] = ub[0]; ud[1] = ub[1]; ud[2] = ub[2]; ud[3] = ub[3];
        }
    }
}

// one wave per row of the pass
__global__ void __launch_bounds__(256) k_batch_scatter(AzScatterArgs a)
{
    __shared__ int off[AZ_BATCH_MAX + 1];
    if (threadIdx.x <= a.n) off[threadIdx.x] = a.off[threadIdx.x];
    __syncthreads();
    const int total = off[a.n];
    const int lane = threadIdx.x & 63;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < total; r += nwaves) {
        int b = 0;
        while (b + 1 < a.n && r >= off[b + 1]) ++b;
        const size_t i = (size_t)(r - off[b]);
        if (lane == 0) a.zoom_d[b][i] = a.zoom[r];
        if (lane < AZ_NSUB) {
            a.score_d[b][i * AZ_NSUB + lane] = a.score[(size_t)r * AZ_NSUB + lane];
            a.keep_d[b][i * AZ_NSUB + lane] = a.keep[(size_t)r * AZ_NSUB + lane];
            if (a.key) a.key_d[b][i * AZ_NSUB + lane] = a.key[(size_t)r * AZ_NSUB + lane];
        }
        if (lane < 4 * AZ_NSUB) a.pred_d[b][i * 4 * AZ_NSUB + lane] = a.pred[(size_t)r * 4 * AZ_NSUB + lane];
    }
}

static_assert(sizeof(AzGatherArgs) <= 4000 && sizeof(AzScatterArgs) <= 4000, "kernel arguments are limited to 4 KB");

}  // namespace

void azk_batch_gather(hipStream_t s, const AzGatherArgs &a)
{
    hipLaunchKernelGGL(k_batch_gather, dim3(16), dim3(256), 0, s, a);
}

void azk_batch_scatter(hipStream_t s, const AzScatterArgs &a)
{
    hipLaunchKernelGGL(k_batch_scatter, dim3(64), dim3(256), 0, s, a);
}
