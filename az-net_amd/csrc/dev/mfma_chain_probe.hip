// dev probe (not part of the library): cycles per MFMA of a DEPENDENT accumulation chain -- NACC independent accumulators
// per wave, WPS waves per SIMD -- for v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32.  One workgroup per CU.
//   hipcc --offload-arch=gfx950 -O3 mfma_chain_probe.hip -o mfma_chain_probe && ./mfma_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(512) k32(float *out, int iters, unsigned long long *cyc)
{
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    floatx16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (s == 1.2345e-30f) out[threadIdx.x] = s;
}

template <int NACC>
__global__ void __launch_bounds__(512) k16(float *out, int iters, unsigned long long *cyc)
{
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    floatx4 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 4; ++e) s += acc[j][e];
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (s == 1.2345e-30f) out[threadIdx.x] = s;
}

template <typename F>
static void run(const char *name, int nacc, int wps, F launch, int iters)
{
    float *out; unsigned long long *cyc, h = 0;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(out, iters, cyc, wps * 256);           // warm-up
    hipEventRecord(e0);
    launch(out, iters, cyc, wps * 256);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * nacc;    // MFMAs per wave
    printf("%-10s acc/wave %d waves/SIMD %d : %7.1f ns per MFMA per wave (%6.1f counter ticks), %7.1f ns per MFMA per SIMD\n",
           name, nacc, wps, ms * 1e6 / n, (double)h / n, ms * 1e6 / (n * wps));
    hipFree(out); hipFree(cyc);
}

int main()
{
    const int iters = 20000;
#define L32(N) [](float *o, int it, unsigned long long *c, int th) { hipLaunchKernelGGL((k32<N>), dim3(256), dim3(th), 0, 0, o, it, c); }
#define L16(N) [](float *o, int it, unsigned long long *c, int th) { hipLaunchKernelGGL((k16<N>), dim3(256), dim3(th), 0, 0, o, it, c); }
    for (int wps = 1; wps <= 2; ++wps) {
        run("32x32x2", 1, wps, L32(1), iters);
        run("32x32x2", 2, wps, L32(2), iters);
        run("32x32x2", 4, wps, L32(4), iters);
        run("16x16x4", 1, wps, L16(1), iters);
        run("16x16x4", 2, wps, L16(2), iters);
        run("16x16x4", 4, wps, L16(4), iters);
        run("16x16x4", 8, wps, L16(8), iters);
    }
    return 0;
}
