// Probe: is v_mfma_f32_16x16x4_f32 bitwise the fmaf chain  c = fmaf(a[k3], b[k3], ... fmaf(a[k0], b[k0], c))
// with the four k slots (lane >> 4) taken in order 0,1,2,3 -- and is that the same chain as two
// v_mfma_f32_32x32x2_f32 (slots 0,1 then the next pair)?  Decides whether a 16-row remainder strip
// could be mixed with 32-row strips without changing a row's bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// C16[16][16] += A[16][K] * Bt[16][K]^T, K = 64, fed 4 k at a time in slot order
__global__ void k16(const float *A, const float *Bt, float *C, int K)
{
    const int l = threadIdx.x, row = l & 15, slot = l >> 4;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[row * K + k0 + slot], Bt[row * K + k0 + slot], acc, 0, 0, 0);
    // C/D layout 16x16: col = lane & 15, row = 4 * (lane >> 4) + e
    for (int e = 0; e < 4; ++e) C[(4 * (l >> 4) + e) * 16 + (l & 15)] = acc[e];
}

// same product on the 32x32x2 instruction (rows/cols 16..31 fed with copies), 2 k at a time
__global__ void k32(const float *A, const float *Bt, float *C, int K)
{
    const int l = threadIdx.x, row = l & 31, slot = l >> 5;
    floatx16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(row & 15) * K + k0 + slot], Bt[(row & 15) * K + k0 + slot], acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) {
        const int r = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), c = l & 31;
        if (r < 16 && c < 16) C[r * 16 + c] = acc[e];
    }
}

int main()
{
    const int K = 64;
    std::vector<float> A(16 * K), Bt(16 * K), C16(256), C32(256), ref(256);
    srand(3);
    for (auto &x : A) x = (rand() / (float)RAND_MAX - 0.5f) * 3.f;
    for (auto &x : Bt) x = (rand() / (float)RAND_MAX - 0.5f) * 0.02f;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float c = 0.f;
            for (int k = 0; k < K; ++k) c = fmaf(A[i * K + k], Bt[j * K + k], c);
            ref[i * 16 + j] = c;
        }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, Bt.size() * 4); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
    hipMemcpy(C16.data(), dC, 256 * 4, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
    hipMemcpy(C32.data(), dC, 256 * 4, hipMemcpyDeviceToHost);
    int d16 = 0, d32 = 0, d1632 = 0;
    double e16 = 0;
    for (int i = 0; i < 256; ++i) {
        d16 += std::memcmp(&C16[i], &ref[i], 4) != 0;
        d32 += std::memcmp(&C32[i], &ref[i], 4) != 0;
        d1632 += std::memcmp(&C16[i], &C32[i], 4) != 0;
        e16 = fmax(e16, fabs((double)C16[i] - ref[i]));
    }
    printf("16x16x4 vs sequential fmaf chain: %d / 256 words differ (max |d| %.3g)\n", d16, e16);
    printf("32x32x2 vs sequential fmaf chain: %d / 256 words differ\n", d32);
    printf("16x16x4 vs 32x32x2:               %d / 256 words differ\n", d1632);
    return 0;
}
