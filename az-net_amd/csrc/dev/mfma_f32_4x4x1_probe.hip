// Probe: is v_mfma_f32_4x4x1_16b_f32 (16 independent 4 x 4 blocks, K = 1 per instruction), issued once per k, bitwise the
// fmaf chain c = fmaf(a[k], b[k], c) in issue order -- and where do its operands and results sit?  Assumed (CDNA ISA):
// lane l = block l / 4; A: row l % 4 of its block, B: column l % 4, D: VGPR v = row v, column l % 4.  Decides whether the
// 8 rows a 40-row launch has past its last full strip could go through this instruction (8 rows x 32 columns per issue as
// 2 row blocks x 8 column blocks) instead of a 16-row half strip, without changing a row's bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));

// out[8][32] = X[8][K] * W[32][K]^T, k in the order the 32x32x2 kernels use inside an 8-wide group: 0,4,1,5,2,6,3,7
__global__ void k4(const float *X, const float *W, float *out, int K)
{
    const int l = threadIdx.x, b = l >> 2, t = l & 3, rb = b >> 3, cb = b & 7;
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    const int ord[8] = {0, 4, 1, 5, 2, 6, 3, 7};
    for (int k0 = 0; k0 < K; k0 += 8)
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + ord[j];
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(X[(rb * 4 + t) * K + k], W[(cb * 4 + t) * K + k], acc, 0, 0, 0);
        }
    for (int v = 0; v < 4; ++v) out[(rb * 4 + v) * 32 + cb * 4 + t] = acc[v];
}

int main()
{
    const int K = 1568;                      // one int6 K chunk
    std::vector<float> X(8 * K), W(32 * K), got(256), ref(256);
    srand(5);
    for (auto &x : X) x = (rand() / (float)RAND_MAX) * 3.f;
    for (auto &x : W) x = (rand() / (float)RAND_MAX - 0.5f) * 0.02f;
    const int ord[8] = {0, 4, 1, 5, 2, 6, 3, 7};
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 32; ++j) {
            float c = 0.f;
            for (int k0 = 0; k0 < K; k0 += 8)
                for (int q = 0; q < 8; ++q) c = fmaf(X[i * K + k0 + ord[q]], W[j * K + k0 + ord[q]], c);
            ref[i * 32 + j] = c;
        }
    float *dX, *dW, *dO;
    hipMalloc(&dX, X.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dO, 256 * 4);
    hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k4, dim3(1), dim3(64), 0, 0, dX, dW, dO, K);
    hipMemcpy(got.data(), dO, 256 * 4, hipMemcpyDeviceToHost);
    int d = 0, dT = 0;
    double e = 0;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 32; ++j) {
            d += std::memcmp(&got[i * 32 + j], &ref[i * 32 + j], 4) != 0;
            e = fmax(e, fabs((double)got[i * 32 + j] - ref[i * 32 + j]));
        }
    // (the other reading of D: VGPR v = column, lane % 4 = row)
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 32; ++j) {
            const int rb = i >> 2, v = i & 3, cb = j >> 2, t = j & 3;
            dT += std::memcmp(&got[(rb * 4 + t) * 32 + cb * 4 + v], &ref[i * 32 + j], 4) != 0;
        }
    printf("4x4x1_16b vs the fmaf chain in k order 0,4,1,5,2,6,3,7 over K = %d: %d / 256 words differ (max |d| %.3g); "
           "with D read transposed: %d / 256\n", K, d, e, dT);
    return 0;
}
