// dev probe (not part of the library): does the 256 MB Infinity Cache serve repeated weight-streaming passes?
// Read-only streams shaped like int6's small-row passes: 512 workgroups, each walks its own contiguous chunk.
//   same:   every pass walks the buffer in the same order (what k_fc_splitk does: LRU over 411 MB > 256 MB -> no hits?)
//   halves: a workgroup walks one chunk of each half; consecutive passes take the halves in alternating order, so a pass
//           starts with the half the previous pass ended with
// build: hipcc --offload-arch=gfx950 -O3 -o mall_probe mall_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 2) k_stream(const f4 *__restrict__ src, size_t n_per_wg_half, size_t half_stride, int first_half,
                                                   int nhalves, float *out)
{
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int hh = 0; hh < nhalves; ++hh) {
        const int h = nhalves == 1 ? 0 : (hh ^ first_half);
        const f4 *p = src + (size_t)h * half_stride + (size_t)blockIdx.x * n_per_wg_half;
        size_t i = threadIdx.x;
        for (; i + 7 * 256 < n_per_wg_half; i += 8 * 256) {
            f4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[i + u * 256];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; i < n_per_wg_half; i += 256) acc += p[i];
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e30f) out[0] = acc.x;
}

int main()
{
    const size_t cap = (size_t)512 << 20;
    f4 *a; float *out;
    hipMalloc(&a, cap); hipMalloc(&out, 64);
    hipMemset(a, 0x3b, cap);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int G = 512;
    auto timed = [&](const char *name, size_t bytes, int nhalves, bool alternate) {
        const size_t n = bytes / 16;
        const size_t per = n / G / nhalves;
        std::vector<float> ms;
        for (int i = 0; i < 24; ++i) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_stream, dim3(G), dim3(256), 0, 0, a, per, per * G, alternate ? (i & 1) : 0, nhalves, out);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1);
            if (i >= 8) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        const float med = ms[ms.size() / 2];
        printf("%-44s %7.1f us  %.2f TB/s\n", name, med * 1e3, per * G * nhalves * 16.0 / (med * 1e-3) / 1e12);
    };
    for (size_t mb : {64, 128, 192, 240, 320, 411}) {
        char nm[96];
        snprintf(nm, 96, "%zu MB, same order every pass", mb); timed(nm, mb << 20, 1, false);
        snprintf(nm, 96, "%zu MB, two halves, same order", mb); timed(nm, mb << 20, 2, false);
        snprintf(nm, 96, "%zu MB, two halves, alternating", mb); timed(nm, mb << 20, 2, true);
    }
    return 0;
}
