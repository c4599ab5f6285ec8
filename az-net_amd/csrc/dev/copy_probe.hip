// dev probe (not part of the library): float4 copy variants through HBM, to pick az_box.hip's form
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
#include <cstdio>
#include <vector>
#include <algorithm>
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_copy(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], &dst[i + u * stride]); else dst[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
int main()
{
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    f4 *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 0x3b, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch) {
        std::vector<float> ms;
        for (int i = 0; i < 8; ++i) { hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1); float t; hipEventElapsedTime(&t, e0, e1); if (i >= 3) ms.push_back(t); }
        std::sort(ms.begin(), ms.end());
        printf("%-28s %.3f ms  %.2f TB/s (r+w)\n", name, ms[ms.size() / 2], 2.0 * bytes / (ms[ms.size() / 2] * 1e-3) / 1e12);
    };
    for (int g : {1024, 2048, 4096, 8192, 16384, 65536}) {
        char nm[64];
        snprintf(nm, 64, "U1 grid %d", g); run(nm, [&]() { hipLaunchKernelGGL((k_copy<1, false>), dim3(g), dim3(256), 0, 0, a, b, n); });
        snprintf(nm, 64, "U4 grid %d", g); run(nm, [&]() { hipLaunchKernelGGL((k_copy<4, false>), dim3(g), dim3(256), 0, 0, a, b, n); });
        snprintf(nm, 64, "U8 grid %d", g); run(nm, [&]() { hipLaunchKernelGGL((k_copy<8, false>), dim3(g), dim3(256), 0, 0, a, b, n); });
        snprintf(nm, 64, "U4 nt grid %d", g); run(nm, [&]() { hipLaunchKernelGGL((k_copy<4, true>), dim3(g), dim3(256), 0, 0, a, b, n); });
    }
    run("hipMemcpyDtoD", [&]() { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
