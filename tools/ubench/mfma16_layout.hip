// Layout check of v_mfma_f32_16x16x32_f16: A lane (i = l & 15, q = l >> 4) elements e <-> A[i][8 q + e]; B likewise B[8 q + e][n = l & 15];
// D lane (n, q) register j <-> D[4 q + j][n].  Prints the maximum deviation from the host product (0 expected).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, i = l & 15, q = l >> 4;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)A[i * 32 + 8 * q + e]; b[e] = (_Float16)B[(8 * q + e) * 16 + i]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) D[(4 * q + j) * 16 + i] = c[j];
}
int main() {
    float hA[16 * 32], hB[32 * 16], hD[256], ref[256];
    for (int i = 0; i < 16 * 32; ++i) hA[i] = (float)((i * 7 + 3) % 13) - 6.f;
    for (int i = 0; i < 32 * 16; ++i) hB[i] = (float)((i * 5 + 1) % 11) - 5.f;
    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) { float s = 0; for (int kk = 0; kk < 32; ++kk) s += hA[r * 32 + kk] * hB[kk * 16 + c]; ref[r * 16 + c] = s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    float m = 0; for (int i = 0; i < 256; ++i) m = fmaxf(m, fabsf(hD[i] - ref[i]));
    printf("max |D - ref| = %g\n", m);
    return 0;
}
