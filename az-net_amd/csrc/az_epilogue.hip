// az_epilogue.hip -- what follows a backbone convolution, in ONE pass over its output: bias + ReLU, and for the layers in
// front of a pooling layer bias + ReLU + 2x2/2 max-pool (ceil mode), on channel-last activations.
// The VGG16 convolutions themselves stay PyTorch-ROCm's (MIOpen), as north_star says; PyTorch then runs the bias add and the
// ReLU as two more element-wise launches over the layer's whole output (and the pool as a third): at 600x1000 that is
// 154 MB read and written twice behind conv1_x -- 0.46 ms per image, 9 % of the CLI's loop (az-net_amd/tools/cli_trace.sh).
// Layer definitions: models/Pascal/VGG16/az-net/test.prototxt:16-384 (Convolution with bias_term, ReLU in place, Pooling MAX
// kernel 2 stride 2; Caffe pools in ceil mode: a last window may be clipped by the map's edge).
// Arithmetic is PyTorch's: fp32 add, max(., 0), max over the window -- and max(relu(y_i + b)) == relu(max(y_i) + b) bit for
// bit for finite values (rounding is monotonic), which is the order used here; NaNs propagate as in torch (relu_nan, max_nan).
#include <hip/hip_runtime.h>

#include "../../include/aznet_hip.h"

namespace {

// NaN-propagating forms, as torch's relu / max_pool2d: a NaN (a broken checkpoint, an overflowed convolution) stays a NaN here
// as it does in the un-fused PyTorch path, instead of being turned into 0 or dropped by a plain comparison.
__device__ __forceinline__ float relu_nan(float v) { return !(v <= 0.f) ? v : 0.f; }
__device__ __forceinline__ float max_nan(float a, float b) { return (a > b || a != a) ? a : b; }

// y: [HW][C] (channel-last), C % 4 == 0.  One float4 per thread and turn.
__global__ void __launch_bounds__(256) k_bias_relu_cl(float4 *__restrict__ y, const float4 *__restrict__ bias, long long n4, int C4)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 v = y[i];
        const float4 b = bias[i % C4];
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        v.x = relu_nan(v.x); v.y = relu_nan(v.y); v.z = relu_nan(v.z); v.w = relu_nan(v.w);
        y[i] = v;
    }
}

// y: [C][HW] (NCHW, PyTorch's default layout -- what tools/prop_az.py runs unless told otherwise)
__global__ void __launch_bounds__(256) k_bias_relu_nchw(float *__restrict__ y, const float *__restrict__ bias, long long n, long long hw, int C)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float v = y[i] + bias[(i / hw) % C];
        y[i] = relu_nan(v);
    }
}

// the same with hw % 4 == 0: a float4 never straddles two channels
__global__ void __launch_bounds__(256) k_bias_relu_nchw4(float4 *__restrict__ y, const float *__restrict__ bias, long long n4, long long hw4, int C)
{
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float b = bias[(i / hw4) % C];
        float4 v = y[i];
        v.x += b; v.y += b; v.z += b; v.w += b;
        v.x = relu_nan(v.x); v.y = relu_nan(v.y); v.z = relu_nan(v.z); v.w = relu_nan(v.w);
        y[i] = v;
    }
}

// y: [C][H][W] -> out: [C][OH][OW]; a thread makes two neighbouring outputs of a row from two float4s (W % 4 == 0) or one
// output from scalars
template <bool VEC>
__global__ void __launch_bounds__(256) k_bias_relu_pool_nchw(const float *__restrict__ y, const float *__restrict__ bias,
                                                             float *__restrict__ out, int C, int H, int W, int OH, int OW)
{
    const long long stride = (long long)gridDim.x * 256;
    if (VEC) {
        const int OW2 = OW >> 1;                               // W % 4 == 0: OW even, pairs of outputs
        const long long n = (long long)C * OH * OW2;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            const int p = (int)(i % OW2);
            const long long r = i / OW2;
            const int oh = (int)(r % OH), c = (int)(r / OH);
            const float *r0 = y + ((long long)c * H + 2 * oh) * W + 4 * p;
            const float4 a = *reinterpret_cast<const float4 *>(r0);
            float m0 = max_nan(a.x, a.y), m1 = max_nan(a.z, a.w);
            if (2 * oh + 1 < H) {
                const float4 b4 = *reinterpret_cast<const float4 *>(r0 + W);
                const float n0 = max_nan(b4.x, b4.y), n1 = max_nan(b4.z, b4.w);
                m0 = max_nan(n0, m0); m1 = max_nan(n1, m1);
            }
            const float b = bias[c];
            m0 += b; m1 += b;
            float2 o;
            o.x = relu_nan(m0); o.y = relu_nan(m1);
            *reinterpret_cast<float2 *>(out + ((long long)c * OH + oh) * OW + 2 * p) = o;
        }
    } else {
        const long long n = (long long)C * OH * OW;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
            const int ow = (int)(i % OW);
            const long long r = i / OW;
            const int oh = (int)(r % OH), c = (int)(r / OH);
            const int h0 = 2 * oh, w0 = 2 * ow;
            const float *r0 = y + ((long long)c * H + h0) * W + w0;
            float m = r0[0];
            const bool h1 = h0 + 1 < H, w1 = w0 + 1 < W;
            if (w1) m = max_nan(r0[1], m);
            if (h1) { m = max_nan(r0[W], m); if (w1) m = max_nan(r0[W + 1], m); }
            m += bias[c];
            out[i] = relu_nan(m);
        }
    }
}

// y: [H][W][C] -> out: [OH][OW][C], OH = ceil(H / 2), OW = ceil(W / 2); window rows / columns past the map are left out
__global__ void __launch_bounds__(256) k_bias_relu_pool_cl(const float4 *__restrict__ y, const float4 *__restrict__ bias,
                                                           float4 *__restrict__ out, int C4, int H, int W, int OH, int OW)
{
    const long long n4 = (long long)OH * OW * C4, stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const int c = (int)(i % C4);
        const long long p = i / C4;
        const int ow = (int)(p % OW), oh = (int)(p / OW);
        const int h0 = 2 * oh, w0 = 2 * ow;
        const bool h1 = h0 + 1 < H, w1 = w0 + 1 < W;
        const float4 *r0 = y + ((long long)h0 * W + w0) * C4 + c;
        float4 m = r0[0];
        auto mx = [&](const float4 &v) {
            m.x = max_nan(v.x, m.x); m.y = max_nan(v.y, m.y); m.z = max_nan(v.z, m.z); m.w = max_nan(v.w, m.w);
        };
        if (w1) mx(r0[C4]);
        if (h1) { mx(r0[(long long)W * C4]); if (w1) mx(r0[(long long)W * C4 + C4]); }
        const float4 b = bias[c];
        m.x += b.x; m.y += b.y; m.z += b.z; m.w += b.w;
        m.x = relu_nan(m.x); m.y = relu_nan(m.y); m.z = relu_nan(m.z); m.w = relu_nan(m.w);
        out[i] = m;
    }
}

int grid_for(long long n)
{
    const long long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" {

int az_bias_relu(void *stream, float *y, const float *bias, int C, long long hw, int channels_last)
{
    if (!y || !bias || C <= 0 || hw < 0) return AZ_ERR_INVALID;
    const long long n = (long long)C * hw;
    if (n == 0) return AZ_OK;
    hipStream_t s = (hipStream_t)stream;
    if (channels_last && (C & 3) == 0 && (((size_t)y | (size_t)bias) & 15) == 0)
        hipLaunchKernelGGL(k_bias_relu_cl, dim3(grid_for(n / 4)), dim3(256), 0, s, (float4 *)y, (const float4 *)bias, n / 4, C / 4);
    else if (channels_last)
        return AZ_ERR_INVALID;         // (a channel count that is no multiple of 4, or unaligned storage: the caller keeps PyTorch's ops)
    else if ((hw & 3) == 0 && ((size_t)y & 15) == 0)
        hipLaunchKernelGGL(k_bias_relu_nchw4, dim3(grid_for(n / 4)), dim3(256), 0, s, (float4 *)y, bias, n / 4, hw / 4, C);
    else
        hipLaunchKernelGGL(k_bias_relu_nchw, dim3(grid_for(n)), dim3(256), 0, s, y, bias, n, hw, C);
    return hipGetLastError() == hipSuccess ? AZ_OK : AZ_ERR_HIP;
}

int az_bias_relu_pool(void *stream, const float *y, const float *bias, float *out, int C, int H, int W, int channels_last)
{
    if (!y || !bias || !out || C <= 0 || H <= 0 || W <= 0) return AZ_ERR_INVALID;
    const int OH = (H + 1) / 2, OW = (W + 1) / 2;
    hipStream_t s = (hipStream_t)stream;
    if (channels_last) {
        if ((C & 3) != 0 || (((size_t)y | (size_t)bias | (size_t)out) & 15) != 0) return AZ_ERR_INVALID;
        hipLaunchKernelGGL(k_bias_relu_pool_cl, dim3(grid_for((long long)OH * OW * (C / 4))), dim3(256), 0, s,
                           (const float4 *)y, (const float4 *)bias, (float4 *)out, C / 4, H, W, OH, OW);
    } else if ((W & 3) == 0 && ((size_t)y & 15) == 0 && ((size_t)out & 7) == 0) {
        hipLaunchKernelGGL((k_bias_relu_pool_nchw<true>), dim3(grid_for((long long)C * OH * (OW / 2))), dim3(256), 0, s, y, bias, out,
                           C, H, W, OH, OW);
    } else {
        hipLaunchKernelGGL((k_bias_relu_pool_nchw<false>), dim3(grid_for((long long)C * OH * OW)), dim3(256), 0, s, y, bias, out,
                           C, H, W, OH, OW);
    }
    return hipGetLastError() == hipSuccess ? AZ_OK : AZ_ERR_HIP;
}

}  // extern "C"
