// Probe: operand/accumulator layout of v_mfma_f32_32x32x16_bf16 and the bf16 split of fp32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned short f2bf(float x)   // round to nearest even
{
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// C[32][32] = A[32][16] * B[16][32]   (A row-major [i][k], Bt row-major [j][k])
__global__ void k(const float *A, const float *Bt, float *C, int parts)
{
    const int l = threadIdx.x;
    const int row = l & 31, kb = (l >> 5) * 8;
    union { bf16x8 v; unsigned short s[8]; } a[3], b[3];
    for (int j = 0; j < 8; ++j) {
        float xa = A[row * 16 + kb + j], xb = Bt[row * 16 + kb + j];
        for (int p = 0; p < 3; ++p) {
            a[p].s[j] = f2bf(xa); xa -= bf2f(a[p].s[j]);
            b[p].s[j] = f2bf(xb); xb -= bf2f(b[p].s[j]);
        }
    }
    floatx16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // order: small terms first
    if (parts == 3) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[1].v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[2].v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2].v, b[0].v, acc, 0, 0, 0);
    }
    if (parts >= 2) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[1].v, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[0].v, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[0].v, acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
        C[r * 32 + (l & 31)] = acc[e];
    }
}

int main()
{
    std::vector<float> A(32 * 16), Bt(32 * 16), C(32 * 32);
    srand(1);
    for (auto &x : A) x = (rand() / (float)RAND_MAX - 0.5f) * 3.f;
    for (auto &x : Bt) x = (rand() / (float)RAND_MAX - 0.5f) * 3.f;
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, Bt.size() * 4); hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
    for (int parts = 1; parts <= 3; ++parts) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, parts);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double r = 0;
                for (int kk = 0; kk < 16; ++kk) r += (double)A[i * 16 + kk] * (double)Bt[j * 16 + kk];
                maxerr = fmax(maxerr, fabs(r - C[i * 32 + j]));
                maxref = fmax(maxref, fabs(r));
            }
        printf("parts=%d max|err|=%.3e (max|ref|=%.2f)\n", parts, maxerr, maxref);
    }
    return 0;
}
