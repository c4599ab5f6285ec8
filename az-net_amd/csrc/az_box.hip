// az_box.hip -- what THIS box sustains: a register-only v_mfma_f32_32x32x2_f32 loop on every SIMD of the chip and a
// float4 copy through HBM, timed with HIP events (az_measure_box).  Boxes of one pool differ by 5-10 % in the clock they
// hold under the matrix pipe; a roofline fraction against the data-sheet peak mixes that into the kernel's figure.  The
// probe's operands are pseudo-random floats, rotated every instruction: a zero-filled loop clocks ~19 % higher than one on
// real data (DVFS, /opt/skills/guides/MI355X_MICROARCH.md) and would overstate what a GEMM can be held to.
#include "az_dev.h"

#include <algorithm>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ float probe_val(unsigned h)
{
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    // random mantissa, exponent of ~2^-7 .. 2^-6, random sign: products ~1e-4, the accumulators stay finite
    return __uint_as_float((h & 0x807FFFFFu) | 0x3C000000u);
}

// One wave per SIMD (256 threads = 4 waves per workgroup, one workgroup per CU and turn), four independent accumulator
// tiles per wave so that the matrix pipe never waits for a result, eight operand pairs taken in rotation.
__global__ void __launch_bounds__(256) k_mfma_probe(float *out, int iters, unsigned seed)
{
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = probe_val(seed + 16u * t + j); b[j] = probe_val(seed + 16u * t + 8 + j); }
    floatx16 acc0, acc1, acc2, acc3;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; acc2[e] = 0.f; acc3[e] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; j += 4) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j + 0], b[j + 0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j + 1], b[j + 1], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j + 2], b[j + 2], acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j + 3], b[j + 3], acc3, 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e] + acc2[e] + acc3[e];
    if (s == 1.2345e-30f) out[t & 1023] = s;          // (never true: keeps the loop's results alive)
}

typedef float f4 __attribute__((ext_vector_type(4)));

// grid-stride float4 copy; U > 1: U loads in flight per thread, non-temporal (streaming) loads and stores
template <int U>
__global__ void __launch_bounds__(256) k_copy_probe(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (U > 1) {
        for (; i + (U - 1) * stride < n; i += U * stride) {
            f4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(&src[i + u * stride]);
#pragma unroll
            for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], &dst[i + u * stride]);
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

double median_ms(std::vector<float> &v)
{
    std::sort(v.begin(), v.end());
    return v.empty() ? 0.0 : (double)v[v.size() / 2];
}

}  // namespace

// mfma_tflops: fp32 MFMA rate the chip holds in a ~3 ms register-only loop (median of 7 launches after 4 warm-ups);
// copy_tbps: (bytes read + bytes written) / time of a float4 copy of copy_bytes (best median of three launch shapes).  Returns 0 or a hipError_t.
int azk_measure_box(hipStream_t s, double *mfma_tflops, double *copy_tbps, size_t copy_bytes)
{
    hipEvent_t ea = nullptr, eb = nullptr;
    hipError_t e;
    if ((e = hipEventCreate(&ea)) != hipSuccess) return (int)e;
    if ((e = hipEventCreate(&eb)) != hipSuccess) { hipEventDestroy(ea); return (int)e; }
    struct Guard { hipEvent_t a, b; void *p0 = nullptr, *p1 = nullptr, *p2 = nullptr;
                   ~Guard() { hipEventDestroy(a); hipEventDestroy(b); if (p0) hipFree(p0); if (p1) hipFree(p1); if (p2) hipFree(p2); } } g{ea, eb};
    int rc = 0;
    auto timed = [&](auto &&launch, std::vector<float> &ms, int warm, int n) {
        for (int i = 0; i < warm + n && !rc; ++i) {
            if (hipEventRecord(ea, s) != hipSuccess) { rc = 1; break; }
            launch();
            if (hipEventRecord(eb, s) != hipSuccess || hipEventSynchronize(eb) != hipSuccess) { rc = 1; break; }
            float t = 0.f;
            if (hipEventElapsedTime(&t, ea, eb) != hipSuccess) { rc = 1; break; }
            if (i >= warm) ms.push_back(t);
        }
    };
    if (mfma_tflops) {
        *mfma_tflops = 0.0;
        if ((e = hipMalloc(&g.p0, 4096)) != hipSuccess) return (int)e;
        int ncu = 256;
        { int dev = 0; hipDeviceProp_t prop; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount; }
        // 8 MFMAs of 64 cycles per iteration and wave: 14 000 iterations ~ 3 ms at 2.4 GHz
        const int iters = 14000, grid = ncu;
        std::vector<float> ms;
        timed([&]() { hipLaunchKernelGGL(k_mfma_probe, dim3(grid), dim3(256), 0, s, (float *)g.p0, iters, 12345u); }, ms, 4, 7);
        if (rc) return (int)hipErrorUnknown;
        const double flop = (double)grid * 4.0 * iters * 8.0 * (32.0 * 32.0 * 2.0 * 2.0);
        *mfma_tflops = flop / (median_ms(ms) * 1e-3) / 1e12;
    }
    if (copy_tbps) {
        *copy_tbps = 0.0;
        const size_t n4 = copy_bytes / 16;
        if (n4 == 0) return 0;
        if ((e = hipMalloc(&g.p1, n4 * 16)) != hipSuccess) return (int)e;
        if ((e = hipMalloc(&g.p2, n4 * 16)) != hipSuccess) return (int)e;
        if ((e = hipMemsetAsync(g.p1, 0x3b, n4 * 16, s)) != hipSuccess) return (int)e;
        // (which launch shape streams best differs a little from box to box -- measured 4.5-5.7 TB/s among these on one
        //  box --: the best median of three forms is the box's figure)
        double best = 1e30;
        for (int form = 0; form < 3 && !rc; ++form) {
            std::vector<float> ms;
            timed([&]() {
                if (form == 0) hipLaunchKernelGGL((k_copy_probe<1>), dim3(1024), dim3(256), 0, s, (const f4 *)g.p1, (f4 *)g.p2, n4);
                else if (form == 1) hipLaunchKernelGGL((k_copy_probe<1>), dim3(65536), dim3(256), 0, s, (const f4 *)g.p1, (f4 *)g.p2, n4);
                else hipLaunchKernelGGL((k_copy_probe<4>), dim3(8192), dim3(256), 0, s, (const f4 *)g.p1, (f4 *)g.p2, n4);
            }, ms, 2, 5);
            const double m = median_ms(ms);
            if (m > 0 && m < best) best = m;
        }
        if (rc || best > 1e29) return (int)hipErrorUnknown;
        *copy_tbps = 2.0 * (double)n4 * 16.0 / (best * 1e-3) / 1e12;
    }
    if (hipGetLastError() != hipSuccess) return (int)hipErrorUnknown;
    return 0;
}
